#!/bin/bash
# usage: bash scripts/band_rows_ab.sh -- band heights 64 / 32 / 16 ("sat.band_rows") at 8 frames per call: the
# band writer's one pass and the two calls (profiles/round5_band_one_pass.txt, step 5)
for br in 64 32 16; do
  echo "== band_rows $br"
  python bench.py --steps 6 --warmup 1 --batch 16 --frames-per-call 8 --no-cpu-baseline --no-variants --no-verify --opt sat.band_rows=$br 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})"
  python bench.py --steps 6 --warmup 1 --batch 16 --frames-per-call 8 --one-pass off --no-cpu-baseline --no-variants --no-verify --opt sat.band_rows=$br 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two calls', d['value'], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})"
done
