#!/bin/bash
# usage: scripts/pmc_s4.sh <tag> [bench args...] -- PMC passes for the sampler study (round 2)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --batch 4 --frames-per-call 1 --no-cpu-baseline"  # one frame per launch: per-launch counters = per frame
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_1 -- $B "$@" > $R/gpurun_out/pmc_${tag}_1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_${tag}_2 -- $B "$@" > $R/gpurun_out/pmc_${tag}_2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_${tag}_3 -- $B "$@" > $R/gpurun_out/pmc_${tag}_3.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_4 -- $B "$@" > $R/gpurun_out/pmc_${tag}_4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_${tag}_5 -- $B "$@" > $R/gpurun_out/pmc_${tag}_5.log 2>&1 || exit 1
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${tag}_1 $R/gpurun_out/pmc_${tag}_2 $R/gpurun_out/pmc_${tag}_3 $R/gpurun_out/pmc_${tag}_4 $R/gpurun_out/pmc_${tag}_5 > $R/gpurun_out/pmc_${tag}_summary.txt
