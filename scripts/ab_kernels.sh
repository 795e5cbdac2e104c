#!/bin/bash
# A/B of library builds on one box with bench_kernels.py: usage scripts/ab_kernels.sh libA.so libB.so ...
for lib in "$@"; do
  F360_LIBRARY=$PWD/foveated-360-video_amd/lib/$lib python bench_kernels.py > gpurun_out/abk_$lib.json || exit 1
  python - "$lib" <<PY
import json, sys
d = json.loads(open("gpurun_out/abk_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k["kernel"].split(" ")[0]: k["us"] for k in d["kernels"]})
PY
done
