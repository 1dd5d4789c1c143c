#!/bin/bash
# HBM traffic of the bench workload's kernels from the PMC counters, one counter per pass (the MI355X guide's recipe:
# separate --pmc runs, no tracing domains beside them).  Run on the GPU box from the repo root:
#   bash tools/collect_pmc.sh [bench args]        -> gpurun_out/pmc_bench_<COUNTER>/ + gpurun_out/pmc_traffic.json
# The program itself follows `--` (python3 bench.py ...), never a wrapper.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="${@:---steps 2 --warmup 1 --no-profile --cpu-seconds 0}"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_bench_$C
  rocprofv3 --pmc $C -d $R/gpurun_out/pmc_bench_$C -o run --output-format csv -- python3 $R/bench.py $ARGS \
    > $R/gpurun_out/pmc_bench_$C.log 2>&1
  echo "[pmc] $C done"
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_bench_FETCH_SIZE $R/gpurun_out/pmc_bench_WRITE_SIZE > $R/gpurun_out/pmc_traffic.json
cat $R/gpurun_out/pmc_traffic.json
