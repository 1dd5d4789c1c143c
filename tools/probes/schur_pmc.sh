#!/bin/bash
# SQ counters of k_schur_blocks inside python bench.py (Venice): where its waves' cycles go
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep -i "LDS\|VALU\|WAIT\|ACTIVE\|INSTS\|BUSY\|WAVE" | tr '\n' ' ' | cut -c1-3000; echo
rm -rf $R/gpurun_out/pmc_schur
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace -d $R/gpurun_out/pmc_schur -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --no-pcg --no-profile --steps 2 --warmup 1 > $R/gpurun_out/pmc_schur.log 2>&1
tail -2 $R/gpurun_out/pmc_schur.log | cut -c1-300
rm -rf $R/gpurun_out/pmc_schur2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE \
  --kernel-trace -d $R/gpurun_out/pmc_schur2 -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --no-pcg --no-profile --steps 2 --warmup 1 > $R/gpurun_out/pmc_schur2.log 2>&1
tail -2 $R/gpurun_out/pmc_schur2.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
for d in ("pmc_schur", "pmc_schur2"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_schur_blocks" not in k and "k_jac_coord" not in k: continue
            k = k.split("(anonymous namespace)::")[-1].split("(")[0][:30]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in tot:
        print(d, k, "launches", len(n[k]), {a: "%.4g" % (b / len(n[k])) for a, b in sorted(tot[k].items())})
PY
find $R/gpurun_out/pmc_schur $R/gpurun_out/pmc_schur2 -name "*.csv" -size +1M -delete 2>/dev/null
