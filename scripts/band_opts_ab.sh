#!/bin/bash
# usage: scripts/band_opts_ab.sh <tag> "<opt list 1>" "<opt list 2>" ... -- the band writer's one
# pass at 8 and at 1 frame per call under different engine options (each list: space-separated
# key=value), two rounds
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_opts_ab.txt
: > $O
for rep in 1 2; do
  for opts in "$@"; do
    o=""; for kv in $opts; do o="$o --opt $kv"; done
    for fpc in 8 1; do
      echo "== rep $rep fpc $fpc [$opts]" >> $O
      python $R/bench.py --steps 8 --warmup 2 --batch 16 --frames-per-call $fpc --one-pass on \
          --no-cpu-baseline --no-variants $o 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['path_hbm_frac_survey_8d'], d['verified']['ok'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})" >> $O || exit 1
    done
  done
done
cat $O
