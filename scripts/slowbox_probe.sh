#!/bin/bash
# On a box whose write path is slow (walker > 86 us per frame) try what might help it; on other
# boxes just report and leave.  One gpurun call.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
one() {
  lib=$1; shift
  F360_LIBRARY=$PWD/foveated-360-video_amd/lib/$lib python bench.py --no-cpu-baseline --no-variants --steps 6 "$@" > gpurun_out/abo.json 2>/dev/null || exit 1
  python - "$lib $*" <<PY
import json, sys
d = json.loads(open("gpurun_out/abo.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["path_hbm_frac"], {k: round(v.get("avg_us_per_frame", v["avg_us"]), 1) for k, v in d["kernels"].items()})
PY
}
one libf360.so
w=$(python -c "import json; d=json.loads(open('gpurun_out/abo.json').read().strip().splitlines()[-1]); print(int(d['kernels']['sat_walk_kernel']['avg_us_per_frame']))")
echo "walker us per frame: $w"
if [ "$w" -lt 86 ]; then echo "not a slow box"; else echo "SLOW BOX"; fi
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "clk\|power" | head -8
for rep in 1 2 3 4 5 6; do
  one libf360.so
  one libf360.so --one-alloc
  one st_sc1.so
  one st_sc1.so --one-alloc
done
