cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/match_time.py 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
for N in 1 2 4; do XFH_LIB_PATH=$PWD/gpurun_exp$N.so python tools/match_time.py 2>&1 | grep -v amdgpu.ids | tail -2 | head -1; done
