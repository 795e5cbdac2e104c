#!/bin/bash
# usage: scripts/prof.sh <tag> [bench args...]   -- rocprofv3 kernel trace + stats, then two PMC passes
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_kt -- python $R/bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}_kt.log 2>&1
echo "kt exit $?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${tag}_pmc1 -- python $R/bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}_pmc1.log 2>&1
echo "pmc1 exit $?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_pmc2 -- python $R/bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}_pmc2.log 2>&1
echo "pmc2 exit $?"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_${tag}_pmc3 -- python $R/bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}_pmc3.log 2>&1
echo "pmc3 exit $?"
cd $R/gpurun_out && find prof_${tag}_* -name "*.csv" | head -30
