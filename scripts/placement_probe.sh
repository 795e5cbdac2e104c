#!/bin/bash
# Table placement against the read-once encoder (profiles/round4_table_placement.txt): the default
# command with one allocation per table, then the write pattern alone (tools/frontbench: build it
# first, see its header).  One gpurun call.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-variants --steps 6 --placement separate > gpurun_out/abo.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/abo.json').read().strip().splitlines()[-1]); print('separate placement:', d['value'], d['path_hbm_frac'], {k: round(v.get('avg_us_per_frame', v['avg_us']), 1) for k, v in d['kernels'].items()})"
timeout -k 10 200 ./tools/frontbench | grep -v "skew"
