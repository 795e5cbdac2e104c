#!/bin/bash
# usage: scripts/pmc_one.sh <tag> <python script relative to the repo> [args...]
# two rocprofv3 --pmc passes (instruction mix / waits) over one script; summarise with
# python scripts/pmc_summary.py gpurun_out/pmc_<tag>_1 gpurun_out/pmc_<tag>_2
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_1 -- python $R/"$@" > $R/gpurun_out/pmc_${tag}_1.log 2>&1
echo "pass1 exit $?"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_${tag}_2 -- python $R/"$@" > $R/gpurun_out/pmc_${tag}_2.log 2>&1
echo "pass2 exit $?"
