#!/bin/bash
# usage: scripts/walk_box_probe.sh -- the read-once kernels with and without their hand-off waits,
# one and two strip owners per SIMD, the shader clock under the walk and the bare write rate of
# the box (boxes of the pool differ by 20 % on the walker)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for o in "sat.walk_variant=1" "sat.walk_variant=2" "debug.ablate=64" "debug.ablate=128" \
         "sat.walk_frames=64" "sat.walk_frames=64 --opt debug.ablate=64" "sat.walk_frames=16"; do
  echo "== $o"
  python $R/scripts/walk_time.py --counts 64 --depths 2 --reps 3 --opt $o 2>/dev/null | grep -v "three kernels"
done
python $R/scripts/walk_time.py --counts 32 --depths 2 --reps 3 2>/dev/null | grep "three kernels"
python $R/scripts/walk_stats.py --frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('launch_us','shader_mhz_median','end_us_min_max')})"
$R/tools/membench 2>&1 | grep -i "write_linear"
(rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -6) || true
