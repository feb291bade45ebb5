#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "rc=$? wall=${SECONDS}s"
tail -3 gpurun_out/bench_default.err | grep -v amdgpu.ids
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "median_of_5x20_steps_fps", "fps_of_5_windows", "public_api_fps", "dense_1024_pairs_per_s", "megadepth1600_pairs_per_s", "lighterglue_frames_per_s", "extraction_only_fps"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us")})
print("roofline_path", d["roofline_path"])
print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")})
for r in d["roofline_kernels"][:12]:
    print(f"  {r['kernel'][:70]:70s} {r['us']:7.1f} us  floor {r['floor_us']:6.1f}  frac {r['frac']:.3f}  {r['bound']}")
PY
