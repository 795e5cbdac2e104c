#!/bin/bash
# usage: scripts_sweep.sh <tag> "<opts A>" "<opts B>" ...   (each opts string is passed to bench.py)
tag=$1; shift
out=gpurun_out/sweep_$tag.jsonl
: > $out
for o in "$@"; do
  echo "## $o" >> $out
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline $o 2>>gpurun_out/sweep_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'kernels':{k:v['avg_us'] for k,v in d['kernels'].items()}}))
" >> $out
done
cat $out
