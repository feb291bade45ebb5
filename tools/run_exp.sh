cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "--- forced dist (RCCL, one rank)"; XFH_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
echo "--- under torchrun, one rank"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
echo "--- --gpus 2 on a one-GPU box"; timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300; echo "rc=${PIPESTATUS[0]}"
echo "--- smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
