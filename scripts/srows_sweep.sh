#!/bin/bash
# rows per wave of the tile streamer, one frame per launch: usage scripts/srows_sweep.sh [bench args]
for n in ${SROWS:-8 5 6 7 9 10 11 13 14 8}; do
  python bench.py --no-cpu-baseline --no-variants --steps 8 --frames-per-call ${FPC:-1} --opt sample.srows=$n "$@" > gpurun_out/sr.json || exit 1
  python - "$n" <<PY
import json, sys
d = json.loads(open("gpurun_out/sr.json").read().strip().splitlines()[-1])
print("srows", sys.argv[1], d["value"], {k: v.get("avg_us_per_frame", v["avg_us"]) for k, v in d["kernels"].items() if "sample" in k})
PY
done
