#!/bin/bash
# source bytes per batched encoder launch ("sat.batch_mb"): usage scripts/batch_mb_sweep.sh [bench args]
for mb in 144 60 100 180 230 300 144; do
  python bench.py --no-cpu-baseline --no-variants --steps 8 --opt sat.batch_mb=$mb "$@" > gpurun_out/bmb.json || exit 1
  python - "$mb" <<PY
import json, sys
d = json.loads(open("gpurun_out/bmb.json").read().strip().splitlines()[-1])
print("sat.batch_mb", sys.argv[1], d["value"], {k: (v.get("frames_per_launch", 1), v.get("avg_us_per_frame", v["avg_us"])) for k, v in d["kernels"].items()})
PY
done
