#!/bin/bash
# usage: scripts/pool_ab.sh <tag> -- what the bound on the table pool costs: the default command's
# set-up with "sat.pool_mb" = automatic (a third of free memory), room for the kept groups plus ONE
# (every losing draw is given back before the next), and the placement experiment of tools/frontbench
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_pool.txt
: > $O
show='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["path_hbm_frac"], d["kernels"]["sat_walk_kernel"]["avg_us_per_frame"], d["config"]["table_placement"]["tried"])'
for rep in 1 2; do
  for mb in 0 46000; do
    echo "== rep $rep sat.pool_mb=$mb (46000: two kept groups + one, at 8K)" >> $O
    python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-verify --opt sat.pool_mb=$mb 2>/dev/null | python -c "$show" >> $O || exit 1
  done
done
echo "== tools/frontbench s (handles mapped in order / shuffled / created round-robin)" >> $O
$R/tools/frontbench s >> $O 2>&1
cat $O
