#!/bin/bash
# usage: scripts/pmc_traffic_only.sh <tag>  -- just the FETCH_SIZE / WRITE_SIZE passes (RGB0 and
# planar source) and profiles-style pmc_traffic.json stamped with the hash of this tree's csrc/
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --batch 4 --frames-per-call 1 --no-cpu-baseline"  # one frame per launch: per-launch counters = per frame
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_4 -- $B > $O/pmc_${tag}_4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_5 -- $B > $O/pmc_${tag}_5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_yuv_4 -- $B --source yuv420p > $O/pmc_${tag}_yuv_4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_yuv_5 -- $B --source yuv420p > $O/pmc_${tag}_yuv_5.log 2>&1 || exit 1
python $R/scripts/pmc_traffic.py $O/${tag}_pmc_traffic.json $O/pmc_${tag}_4 $O/pmc_${tag}_5 $O/pmc_${tag}_yuv_4 $O/pmc_${tag}_yuv_5
