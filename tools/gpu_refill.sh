cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/bench_src/mfma_refill.hip -o /tmp/mfma_refill 2>/dev/null && timeout 200 /tmp/mfma_refill 2.5 | tee gpurun_out/r04_mfma_refill.txt | cut -c1-260
V1=$(python -c "print(','.join(str(1000+i) for i in range(16)))")
XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_scan.so timeout 300 python tools/head_soak.py --variants $V1 --foreign none --iters 100000000 --max-seconds 2.5 --logits 0 2>&1 | grep "^variant.*foreign" | sort -u | tee gpurun_out/r04_mfma_refill_control.txt | awk '{print $2, $5, $13}' | tr '\n' ';'
