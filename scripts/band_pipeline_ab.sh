#!/bin/bash
# usage: scripts/band_pipeline_ab.sh <tag> -- the band writer's one pass and the two calls at 8
# frames per call: one stream, launch groups alternating over two streams ("sat.pipeline" 1),
# reducers + carry passes on the side stream and writers on the context's (2)
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_pipeline_ab.txt
: > $O
for rep in 1 2; do
for pl in 1 2 0; do
  for mode in auto off; do
    echo "== rep $rep sat.pipeline=$pl one-pass $mode" >> $O
    python $R/bench.py --steps 8 --warmup 2 --batch 16 --frames-per-call 8 --one-pass $mode \
        --no-cpu-baseline --no-variants --opt sat.pipeline=$pl 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['path_hbm_frac_survey_8d'], d['verified']['ok'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})" >> $O || exit 1
  done
done
done
cat $O
