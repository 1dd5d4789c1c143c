#!/bin/bash
# usage: sweep_env.sh VAR "v1 v2 ..." workload kernel_class [steps]  -- time one kernel class of bench.py under values of an env var
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VAR=$1; VALS=$2; W=$3; K=$4; N=${5:-10}
for v in $VALS; do
  env $VAR=$v python bench.py --workload $W --steps $N --warmup 3 --cpu-seconds 0 --cpu-full none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$W', '$K ms/launch %.3f' % (d['kernel_ms']['$K']/$N), 'ms_per_step %.3f' % d['ms_per_step'])"
done
