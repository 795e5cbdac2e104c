#!/bin/bash
# usage: scripts/pmc_py.sh <tag> <python script relative to the repo> [args...]
# counter passes (instruction mix, waits, LDS, HBM traffic) over one script; the summary lands in
# gpurun_out/pmc_<tag>_summary.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${tag}_1 -- python $R/"$@" > $O/pmc_${tag}_1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_${tag}_2 -- python $R/"$@" > $O/pmc_${tag}_2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_4 -- python $R/"$@" > $O/pmc_${tag}_4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${tag}_5 -- python $R/"$@" > $O/pmc_${tag}_5.log 2>&1 || exit 1
rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/pmc_${tag}_6 -- python $R/"$@" > $O/pmc_${tag}_6.log 2>&1 || exit 1
python $R/scripts/pmc_summary.py $O/pmc_${tag}_1 $O/pmc_${tag}_2 $O/pmc_${tag}_4 $O/pmc_${tag}_5 $O/pmc_${tag}_6 > $O/pmc_${tag}_summary.txt
