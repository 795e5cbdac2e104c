#!/bin/bash
# usage: scripts/band_sizes_ab.sh <tag> -- one pass (auto) against the two calls (off) where the
# band writer takes the call: other frame counts at 8K, other frame sizes
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_sizes_ab.txt
: > $O
run() {  # width height batch frames-per-call
  for mode in auto off; do
    echo "== $1x$2 batch $3, $4 frames per call, one-pass $mode" >> $O
    python $R/bench.py --width $1 --height $2 --batch $3 --frames-per-call $4 --one-pass $mode --steps 6 --warmup 2 \
        --no-cpu-baseline --no-variants 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['path_hbm_frac_survey_8d'], d['verified']['ok'], d['config']['encoder'][:40], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})" >> $O || exit 1
  done
}
run 7680 3840 16 2
run 7680 3840 16 4
run 7680 3840 32 16
run 7680 3840 44 22
run 3840 1920 32 8
run 3840 1920 64 32
run 1920 1080 64 16
run 1920 1080 128 64
run 2560 1440 64 16
cat $O
