#!/bin/bash
# SQ counters of the update micro-benchmark's variants (full-entropy operands): where the waves' cycles go.
#   bash tools/pmc_update_probe.sh "0,128,1,129"   -> gpurun_out/pmc_update_probe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-0,128}
cd /tmp && export TMPDIR=/tmp
export BA_BENCH_NONZERO=1 BA_BENCH_ENTROPY=1 BA_BENCH_REPS=10
rm -rf $R/gpurun_out/pmc_upd
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE \
  --kernel-trace -d $R/gpurun_out/pmc_upd -o run --output-format csv -- python3 $R/tools/bench_update.py 126 $V > $R/gpurun_out/pmc_upd.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$R/gpurun_out/pmc_upd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_ldl_update" not in k: continue
        k = k.split("(anonymous namespace)::")[-1].split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_upd/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_ldl_update" not in k: continue
        k = k.split("(anonymous namespace)::")[-1].split("(")[0][:60]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
with open("$R/gpurun_out/pmc_update_probe.txt", "w") as o:
    for k in tot:
        d = len(n[k]); c = {a: b / d for a, b in tot[k].items()}
        us = sorted(dur[k])[len(dur[k]) // 2]
        line = f"{k}: launches {d} median {us:.0f} us; " + " ".join(f"{a}={b:.4g}" for a, b in sorted(c.items()))
        if "GRBM_GUI_ACTIVE" in c: line += f" | clock {c['GRBM_GUI_ACTIVE'] / 8 / us / 1e3:.3f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c: line += f" mfma_busy/sq_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}"
        print(line); o.write(line + "\n")
PY
find $R/gpurun_out/pmc_upd -name "*.csv" -size +2M -delete
