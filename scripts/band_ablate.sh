#!/bin/bash
# usage: scripts/band_ablate.sh <tag> -- where the band writer's one pass spends its time: the same
# command with parts of the emit switched off (debug.ablate: 1024 no emit at all, 2048 no
# one-column boxes, 4096 no wide boxes / side rows; results are wrong, hence --no-verify)
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_band_ablate.txt
: > $O
for ab in ${ABLATES:-0 1024 2048 4096 6144 8192}; do
  for fpc in ${FPCS:-1 8}; do
    echo "== debug.ablate=$ab frames-per-call $fpc" >> $O
    python $R/bench.py --steps 6 --warmup 1 --batch 16 --frames-per-call $fpc --one-pass on \
        --no-cpu-baseline --no-variants --no-verify --opt debug.ablate=$ab 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})" >> $O || exit 1
  done
done
cat $O
