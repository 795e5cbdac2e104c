#!/bin/bash
# usage: scripts/round_profile.sh <tag>   -- everything profiles/ holds for one state of the tree
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/${tag}_bench_default.json || exit 1
python $R/bench.py --streams 3 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_streams3.json || exit 1
python $R/bench.py --source yuv420p --no-cpu-baseline > $R/gpurun_out/${tag}_bench_yuv420p.json || exit 1
python $R/bench.py --fused --no-cpu-baseline > $R/gpurun_out/${tag}_bench_fused.json || exit 1
python $R/bench.py --fused --source yuv420p --no-cpu-baseline > $R/gpurun_out/${tag}_bench_fused_yuv420p.json || exit 1
python $R/bench.py --fused --source yuv420p --streams 3 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_fused_yuv420p_streams3.json || exit 1
python $R/bench_kernels.py > $R/gpurun_out/${tag}_bench_kernels_8k.json || exit 1
# the default command itself under the kernel tracer: its JSON line and the tracer's per-kernel
# averages come from the same run
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_default -- python $R/bench.py > $R/gpurun_out/${tag}_bench_default_under_rocprof.json 2> $R/gpurun_out/prof_${tag}_default.log) || exit 1
$R/scripts/prof.sh ${tag} || exit 1
$R/scripts/prof.sh ${tag}_yuv --source yuv420p || exit 1
