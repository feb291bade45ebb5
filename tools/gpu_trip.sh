cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python tools/ab_option.py bx 21 23 111 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bx23_ab_block41.log
timeout 200 python tools/ab_option.py bx 21 23 112 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bx23_ab_block42.log
