#!/bin/bash
# rocprofv3 kernel stats of a short bench run, filtered:  bash tools/prof_kernels.sh <tag> <name-pattern> [bench args...]
TAG=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
grep -E "$PAT" $f | cut -c1-60,150-260
