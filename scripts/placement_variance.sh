#!/bin/bash
# usage: scripts/placement_variance.sh <tag> -- run-to-run spread of the default command on one box,
# with the caller's frames and reduced frames as one allocation each (slab, the default) or one per frame
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_placement_variance.txt
: > $O
for rep in 1 2 3 4; do
  for fp in slab separate; do
    echo "== rep $rep --frame-placement $fp" >> $O
    python $R/bench.py --frame-placement $fp --no-cpu-baseline --no-variants 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['path_hbm_frac'], d['kernels']['sat_walk_kernel']['avg_us_per_frame'], d['config']['table_placement']['tried'][0].split(': ',1)[1])" >> $O || exit 1
  done
done
cat $O
