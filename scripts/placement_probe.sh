#!/bin/bash
# bench.py's placement calibration (profiles/round4_table_placement.txt) against a fixed placement,
# alternating; one gpurun call.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
one() {
  python bench.py --no-cpu-baseline --no-variants "$@" > gpurun_out/abo.json 2> gpurun_out/abo.err || { tail -5 gpurun_out/abo.err; return; }
  python - "$*" <<PY
import json, sys
d = json.loads(open("gpurun_out/abo.json").read().strip().splitlines()[-1])
p = d["config"]["table_placement"]
print(sys.argv[1], d["value"], d["path_hbm_frac"], {k: round(v.get("avg_us_per_frame", v["avg_us"]), 1) for k, v in d["kernels"].items()},
      p["chosen"], p["tried"])
PY
}
for rep in 1 2 3; do
  one --placement separate
  one
done
