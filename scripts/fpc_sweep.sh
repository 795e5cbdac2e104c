#!/bin/bash
# frames per call of the batched two-call path: usage scripts/fpc_sweep.sh [bench args...]
for round in 1 2; do
  for n in 1 2 4 8 16; do
    python bench.py --no-cpu-baseline --no-variants --steps 8 --frames-per-call $n "$@" > gpurun_out/fpc.json || exit 1
    python - "$n" <<PY
import json, sys
d = json.loads(open("gpurun_out/fpc.json").read().strip().splitlines()[-1])
print("frames_per_call", sys.argv[1], d["value"], d["path_hbm_frac"], d["roofline"]["frac"], {k: v.get("avg_us_per_frame", v["avg_us"]) for k, v in d["kernels"].items()})
PY
  done
done
