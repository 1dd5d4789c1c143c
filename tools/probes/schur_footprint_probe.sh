cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for fp in 0 4096 262144; do
  rm -rf $R/gpurun_out/sp_$fp
  if [ $fp = 0 ]; then unset BA_SCHUR_PROBE_FOOTPRINT; else export BA_SCHUR_PROBE_FOOTPRINT=$fp; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sp_$fp -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --no-pcg --no-profile --steps 2 --warmup 1 > $R/gpurun_out/sp_$fp.log 2>&1
  f=$(find $R/gpurun_out/sp_$fp -name "*kernel_stats.csv" | head -1)
  echo "footprint $fp:"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n = r['Name']
    if any(k in n for k in ('k_schur_blocks', 'k_schur_chunks', 'k_obs_y')): print('  ', n.split('::')[-1][:24], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
"
  rm -rf $R/gpurun_out/sp_$fp
done
