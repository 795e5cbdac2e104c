#!/bin/bash
# usage: scripts/walk_ablate.sh  -- where the strip walker's time goes (8K, 64 frames per launch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for o in 0 64 128 192; do
  echo "debug.ablate=$o"
  python $R/scripts/walk_time.py --counts 64 --depths 2 --reps 3 --opt debug.ablate=$o | grep walk
done
