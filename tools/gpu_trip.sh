cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/fx_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fx_check.log
timeout 200 python tools/ab_option.py fx 0 1 117 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fx_ab.log
timeout 600 python tools/head_soak.py --variants 10 --foreign backbone --iters 100000000 --max-seconds 80 2>&1 | grep -v amdgpu.ids | tee gpurun_out/head_soak_t3_v10.log
timeout 600 python tools/head_soak.py --variants 0,11,13 --foreign backbone --iters 100000000 --max-seconds 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/head_soak_t3.log
