cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "star or dense or fine or match_many" 2>&1 | tail -3
timeout 200 python bench.py --workload dense --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
python tools/gpu_fine_time.py 100000 2>&1 | grep fine_matcher
