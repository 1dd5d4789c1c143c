#!/bin/bash
# Round-end evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
#   bash tools/round_profiles.sh <tag>      -> gpurun_out/<tag>_*   (copy what should be judged into profiles/)
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_venice.json 2> gpurun_out/${TAG}_bench_venice.err
tail -c 600 gpurun_out/${TAG}_bench_venice.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 \
  > $R/gpurun_out/${TAG}_prof.log 2>&1
find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats_venice_bench.csv \;
find $R/gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete   # large; the stats summary is what is kept
head -12 $R/gpurun_out/${TAG}_kernel_stats_venice_bench.csv
cd $R && bash tools/collect_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1 && cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
rm -rf gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
echo "[round_profiles] done"
