#!/bin/bash
# usage: bash scripts/band_yuv_ab.sh -- planar YUV 4:2:0 sources at 8 frames per call: the band writer's one pass
# (auto) against the two planar calls (off) (profiles/round5_band_one_pass.txt, step 13)
for rep in 1 2; do for mode in auto off; do
echo "== rep $rep yuv420p 8 frames per call one-pass $mode"
python bench.py --source yuv420p --steps 8 --warmup 2 --batch 16 --frames-per-call 8 --one-pass $mode --no-cpu-baseline --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['path_hbm_frac_survey_8d'], d['verified']['ok'], d['config']['encoder'][:40], {k:(v.get('avg_us_per_frame',v['avg_us'])) for k,v in d['kernels'].items()})"
done; done
