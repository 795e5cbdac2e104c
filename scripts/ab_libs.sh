#!/bin/bash
# A/B of library builds on one box, three alternating rounds of the default command (20 steps):
# usage scripts/ab_libs3.sh "<bench args>" libA.so libB.so ...
args="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do
    F360_LIBRARY=$PWD/foveated-360-video_amd/lib/$lib python bench.py --no-cpu-baseline --no-variants $args > gpurun_out/ab_${lib}_$round.json || exit 1
    python - "$lib" "$round" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d.get("path_hbm_frac"), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
  done
done
