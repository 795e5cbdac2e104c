#!/bin/bash
# usage: scripts/round_profile.sh <tag>   -- everything profiles/ holds for one state of the tree
# (bench lines of the variants, the default command under the kernel tracer, PMC passes for the
# RGB0 and the planar source, profiles/pmc_traffic.json stamped with the hash of csrc/)
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py > $O/${tag}_bench_default.json || exit 1
python $R/bench.py --frames-per-call 1 --no-cpu-baseline > $O/${tag}_bench_per_frame_calls.json || exit 1
python $R/bench.py --streams 3 --no-cpu-baseline > $O/${tag}_bench_streams3.json || exit 1
python $R/bench.py --source yuv420p --no-cpu-baseline > $O/${tag}_bench_yuv420p.json || exit 1
python $R/bench.py --fused --no-cpu-baseline > $O/${tag}_bench_fused.json || exit 1
python $R/bench.py --fused --source yuv420p --no-cpu-baseline > $O/${tag}_bench_fused_yuv420p.json || exit 1
python $R/bench.py --fused --source yuv420p --streams 3 --no-cpu-baseline > $O/${tag}_bench_fused_yuv420p_streams3.json || exit 1
python $R/bench_kernels.py > $O/${tag}_bench_kernels_8k.json || exit 1
python $R/tests/bench_configs.py --config all > $O/${tag}_bench_configs.jsonl 2> $O/${tag}_bench_configs.err || exit 1
echo "bench lines done"
# the default command itself under the kernel tracer: its JSON line and the tracer's per-kernel
# averages come from the same run
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_default -- python $R/bench.py > $O/${tag}_bench_default_under_rocprof.json 2> $O/prof_${tag}_default.log) || exit 1
echo "kernel trace done"
$R/scripts/pmc_s4.sh ${tag} || exit 1
echo "pmc rgb0 done"
$R/scripts/pmc_s4.sh ${tag}_yuv --source yuv420p || exit 1
echo "pmc yuv done"
python $R/scripts/pmc_traffic.py $O/${tag}_pmc_traffic.json $O/pmc_${tag}_4 $O/pmc_${tag}_5 $O/pmc_${tag}_yuv_4 $O/pmc_${tag}_yuv_5 || exit 1
# the tracer's per-kernel table, small enough to keep
find $O/prof_${tag}_default -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats_default_command.csv \;
