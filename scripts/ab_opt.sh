#!/bin/bash
# A/B of one engine option on one box: usage scripts/ab_opt.sh <key> "<values>" [bench args...]
key=$1; vals=$2; shift 2
for round in 1 2; do
  for v in $vals; do
    python bench.py --no-cpu-baseline --no-variants --steps 8 --opt $key=$v "$@" > gpurun_out/abo.json || exit 1
    python - "$key=$v" <<PY
import json, sys
d = json.loads(open("gpurun_out/abo.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
  done
done
