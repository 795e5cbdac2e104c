#!/bin/bash
# usage: scripts/pmc_traffic_only.sh <tag>  -- the FETCH_SIZE / WRITE_SIZE passes (RGB0 and planar
# source) of the DEFAULT command's kernels (64 frames per encoder launch, 16 per sampler launch)
# and of the one-frame-per-call kernels, and a profiles-style pmc_traffic.json stamped with the
# hash of this tree's csrc/.  Counters only: no tracing domain beside --pmc.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants"                   # the default command (one pass), shortened
T="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --one-pass off"    # the two calls
B="python $R/bench.py --steps 2 --warmup 1 --batch 4 --frames-per-call 1 --no-cpu-baseline --no-variants"  # one frame per launch
E="python $R/bench.py --steps 2 --warmup 1 --batch 8 --frames-per-call 8 --no-cpu-baseline --no-variants --opt sat.pipeline=0"  # the band writer's one pass (one stream: a launch's counters are its own)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_d4 -- $D > $O/pmc_${tag}_d4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_d5 -- $D > $O/pmc_${tag}_d5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_t4 -- $T > $O/pmc_${tag}_t4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_t5 -- $T > $O/pmc_${tag}_t5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_4 -- $B > $O/pmc_${tag}_4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_5 -- $B > $O/pmc_${tag}_5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_e4 -- $E > $O/pmc_${tag}_e4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_e5 -- $E > $O/pmc_${tag}_e5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_yuv_d4 -- $D --source yuv420p > $O/pmc_${tag}_yuv_d4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_yuv_d5 -- $D --source yuv420p > $O/pmc_${tag}_yuv_d5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_yuv_t4 -- $T --source yuv420p > $O/pmc_${tag}_yuv_t4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_yuv_t5 -- $T --source yuv420p > $O/pmc_${tag}_yuv_t5.log 2>&1 || exit 1
python $R/scripts/pmc_traffic.py $O/${tag}_pmc_traffic.json $O/pmc_${tag}_d4:$O/pmc_${tag}_t4:$O/pmc_${tag}_4 $O/pmc_${tag}_d5:$O/pmc_${tag}_t5:$O/pmc_${tag}_5 $O/pmc_${tag}_yuv_d4:$O/pmc_${tag}_yuv_t4 $O/pmc_${tag}_yuv_d5:$O/pmc_${tag}_yuv_t5 $O/pmc_${tag}_e4 $O/pmc_${tag}_e5
