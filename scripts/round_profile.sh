#!/bin/bash
# usage: scripts/round_profile.sh <tag>   -- everything profiles/ holds for one state of the tree
# (bench lines of the default command and its variants, the default command under the kernel
# tracer, PMC passes of the default command's kernels for the RGB0 and the planar source,
# profiles/pmc_traffic.json stamped with the hash of csrc/).  Counters and tracing never share a
# run.  A progress line after every step.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py > $O/${tag}_bench_default.json || exit 1
echo "default done"
python $R/bench.py --one-pass off --no-cpu-baseline --no-variants > $O/${tag}_bench_two_calls.json || exit 1
python $R/bench.py --frames-per-call 1 --no-cpu-baseline --no-variants > $O/${tag}_bench_per_frame_calls.json || exit 1
python $R/bench.py --frames-per-call 8 --no-cpu-baseline > $O/${tag}_bench_8_frames_per_call_band_one_pass.json || exit 1
python $R/bench.py --frames-per-call 8 --one-pass off --no-cpu-baseline --no-variants > $O/${tag}_bench_8_frames_per_call_two_calls.json || exit 1
python $R/bench.py --frames-per-call 1 --one-pass on --opt fuse.band=2 --no-cpu-baseline --no-variants > $O/${tag}_bench_1_frame_per_call_band_one_pass.json || exit 1
python $R/bench.py --frames-per-call 32 --no-cpu-baseline --no-variants > $O/${tag}_bench_32_frames_per_call.json || exit 1
python $R/bench.py --streams 2 --no-cpu-baseline --no-variants > $O/${tag}_bench_streams2.json || exit 1
python $R/bench.py --source yuv420p --no-cpu-baseline --no-variants > $O/${tag}_bench_yuv420p.json || exit 1
python $R/bench.py --fused --no-cpu-baseline > $O/${tag}_bench_fused.json || exit 1
python $R/bench.py --fused --source yuv420p --no-cpu-baseline > $O/${tag}_bench_fused_yuv420p.json || exit 1
python $R/bench.py --global-batch 64 --no-cpu-baseline --no-variants > $O/${tag}_bench_global_batch_64_n1.json || exit 1
echo "bench variants done"
python $R/bench_kernels.py > $O/${tag}_bench_kernels_8k.json || exit 1
python $R/bench_kernels.py --width 3840 --height 1920 > $O/${tag}_bench_kernels_3840x1920.json || exit 1
echo "bench_kernels done"
python $R/tests/bench_configs.py --config all > $O/${tag}_bench_configs.jsonl 2> $O/${tag}_bench_configs.err || exit 1
echo "bench_configs done"
# the default command itself under the kernel tracer: its JSON line and the tracer's per-kernel
# averages come from the same run
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_default -- python $R/bench.py > $O/${tag}_bench_default_under_rocprof.json 2> $O/prof_${tag}_default.log) || exit 1
find $O/prof_${tag}_default -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_default_command.csv \;
echo "kernel trace done"
# PMC: the default command, shortened (2 steps), five counter groups, one run each
cd /tmp && export TMPDIR=/tmp
D="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${tag}_1 -- $D > $O/pmc_${tag}_1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_${tag}_2 -- $D > $O/pmc_${tag}_2.log 2>&1 || exit 1
echo "pmc groups 1-2 done"
cd $R && $R/scripts/pmc_traffic_only.sh ${tag} || exit 1
echo "pmc traffic done"
# the sibling kernels' HBM-side bytes (bench_kernels.py, 3 repetitions): FETCH_SIZE / WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
K="python $R/bench_kernels.py --reps 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_k4 -- $K > $O/pmc_${tag}_k4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${tag}_k5 -- $K > $O/pmc_${tag}_k5.log 2>&1 || exit 1
python $R/scripts/pmc_summary.py $O/pmc_${tag}_k4 $O/pmc_${tag}_k5 > $O/${tag}_siblings_pmc_summary.txt
echo "sibling pmc done"
cd $R
python $R/scripts/pmc_summary.py $O/pmc_${tag}_1 $O/pmc_${tag}_2 $O/pmc_${tag}_d4 $O/pmc_${tag}_d5 $O/pmc_${tag}_t4 $O/pmc_${tag}_t5 > $O/${tag}_pmc_summary.txt
python $R/scripts/pmc_summary.py $O/pmc_${tag}_yuv_d4 $O/pmc_${tag}_yuv_d5 > $O/${tag}_yuv_pmc_summary.txt
echo "all done"
