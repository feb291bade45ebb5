cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
V1=$(python -c "print(','.join(str(1000+i) for i in range(16)))")
V3=$(python -c "print(','.join(str(3000+i) for i in range(16)))")
XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_scan.so timeout 500 python tools/head_soak.py --variants $V1,$V3 --foreign none --iters 100000000 --max-seconds 4 --logits 0 2>&1 | grep "^variant.*foreign" | tee gpurun_out/r04_head_scan_same_box.txt | cut -c1-150
