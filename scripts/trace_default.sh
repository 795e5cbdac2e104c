#!/bin/bash
# the default bench command under the kernel tracer (its JSON line and the tracer's averages from one run)
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_default -- python $R/bench.py > $R/gpurun_out/${tag}_bench_default_under_rocprof.json 2> $R/gpurun_out/prof_${tag}_default.log
