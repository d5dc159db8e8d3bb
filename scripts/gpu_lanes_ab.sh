#!/bin/bash
# A/B of the number of lanes nova_hip_render_samples rotates over (run on the GPU box).
# usage: gpu_lanes_ab.sh <outdir>
set -u
out=${1:-gpurun_out/lanes}
mkdir -p $out
for l in 1 2 3 4; do
  NOVA_HIP_LANES=$l timeout -k 10 200 python3 -u bench.py --no-cpu-baseline > $out/bench_l$l.json 2> $out/bench_l$l.err || { echo "lanes $l failed"; tail -5 $out/bench_l$l.err; exit 1; }
  NOVA_HIP_LANES=$l timeout -k 10 200 python3 -u -m pytest tests/test_gpu_render_samples.py -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest_l$l.log 2>&1 || { echo "tests with $l lanes failed"; tail -20 $out/pytest_l$l.log; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("$out/bench_l$l.json").read().strip().splitlines()[-1])
print("lanes $l: four-call schedule", d["ms_per_step"], "ms; fused", d["fused_samples"], open("$out/pytest_l$l.log").read().strip().splitlines()[-1])
PY
done
