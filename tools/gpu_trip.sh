cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "alternative or invnorm or fused_heat or small" > gpurun_out/pytest_heads.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_heads.log
timeout 200 python tools/ab_option.py heads_f32 1 2 203 2>&1 | grep -v amdgpu.ids | tee gpurun_out/heads_ab_kp.log
timeout 200 python tools/ab_option.py heads_f32 1 2 202 2>&1 | grep -v amdgpu.ids | tee gpurun_out/heads_ab_rel.log
V3=$(python -c "print(','.join(str(3000+i) for i in range(16)))")
timeout 600 python tools/head_soak.py --variants $V3 --foreign none --iters 100000000 --max-seconds 6 --logits 0 2>&1 | grep -v amdgpu.ids | grep "^variant.*foreign" | tee gpurun_out/head_soak_shift_f32r.log
