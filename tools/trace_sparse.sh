#!/bin/bash
# rocprofv3 kernel trace of a short bench run, one factorisation's timeline printed:  bash tools/trace_sparse.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_$TAG -o run --output-format rocpd csv -- python3 $R/bench.py --cpu-seconds 0 --no-pcg --no-profile "$@" > $R/gpurun_out/trace_$TAG.log 2>&1
db=$(find $R/gpurun_out/trace_$TAG -name "*.db" | head -1)
python3 $R/tools/trace_timeline.py $db 60 > $R/gpurun_out/trace_${TAG}_timeline.txt 2>&1
f=$(find $R/gpurun_out/trace_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/trace_${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/trace_$TAG
