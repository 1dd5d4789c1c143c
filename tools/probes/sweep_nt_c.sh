#!/bin/bash
# the bulk update's C tiles read / written non-temporally (BA_LDL_NT_C=1) against the default, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/bundleadjustment.jl_amd/csrc
COMMON="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-local-typedef"
for nt in 1 0 1 0; do
  /opt/rocm/bin/hipcc $COMMON -DBA_LDL_NT_C=$nt -c ba_dense_ldl.hip -o ba_dense_ldl.o && \
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libba_hip.so ba_api.o ba_model_kernels.o ba_normal_kernels.o ba_dense_ldl.o ba_lm.o ba_comm.o ba_bal_reader.o ba_order.o -ldl
  cd $R
  python bench.py --cpu-seconds 0 --cpu-full none --no-pcg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NT_C', $nt, 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg launch ms', d['roofline']['avg_launch_ms'])"
  cd $R/bundleadjustment.jl_amd/csrc
done
