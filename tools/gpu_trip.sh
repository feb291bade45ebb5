cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids
timeout 300 python tools/fx_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fx_check2.log; tail -22 gpurun_out/fx_check2.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_t11.log; tail -c 3000 gpurun_out/bench_t11.log
