cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "adversarial or filter_and_refine or match or sizes_beyond" 2>&1 | tail -3
python tools/match_time.py 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print({k: d.get(k) for k in ('value', 'median_of_5x20_steps_fps', 'public_api_fps', 'dense_1024_pairs_per_s', 'megadepth1600_pairs_per_s', 'lighterglue_frames_per_s')})"
