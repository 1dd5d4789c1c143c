#!/bin/bash
# rebuild ba_normal_kernels with different prefetch depths of k_schur_blocks and time the kernel class on Venice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/bundleadjustment.jl_amd/csrc
COMMON="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-local-typedef"
for pf in "$@"; do
  /opt/rocm/bin/hipcc $COMMON -DBA_SCHUR_PF=$pf -c ba_normal_kernels.hip -o ba_normal_kernels.o && \
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libba_hip.so ba_api.o ba_model_kernels.o ba_normal_kernels.o ba_dense_ldl.o ba_lm.o ba_comm.o ba_bal_reader.o ba_order.o -ldl
  cd $R
  python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --cpu-full none --no-pcg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PF', $pf, 'schur_blocks ms/launch', d['kernel_ms']['k_schur_blocks']/10, 'ms_per_step', d['ms_per_step'])"
  cd $R/bundleadjustment.jl_amd/csrc
done
