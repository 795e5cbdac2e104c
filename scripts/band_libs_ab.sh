#!/bin/bash
# usage: scripts/band_libs_ab.sh <tag> lib1.so lib2.so ... -- variant builds of the library
# (make -C foveated-360-video_amd/csrc OUT=../lib/<name>.so BUILD=build_<name> EXTRA=-D...) at 8
# frames per call, pipelined and on one stream, two rounds
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${tag}_libs_ab.txt
: > $O
for rep in 1 2; do
  for lib in "$@"; do
    for pl in 1 0; do
      echo "== rep $rep $lib sat.pipeline=$pl" >> $O
      F360_LIBRARY=$R/foveated-360-video_amd/lib/$lib python $R/bench.py --steps 8 --warmup 2 --batch 16 --frames-per-call 8 \
          --no-cpu-baseline --no-variants --opt sat.pipeline=$pl 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['verified']['ok'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})" >> $O || exit 1
    done
  done
done
cat $O
