#!/bin/bash
# MFMA-pipe utilisation and effective clock of the factorisation kernels INSIDE `python bench.py` (Venice shape):
# one rocprofv3 --pmc pass (SQ + GRBM counters) with the kernel trace.  Under --pmc the profiler serialises kernels, so the
# factorisation runs its in-order schedule (it falls back by itself).   -> gpurun_out/pmc_mfma_bench.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mfma
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace -d $R/gpurun_out/pmc_mfma -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --no-pcg --no-profile --steps 4 --warmup 1 > $R/gpurun_out/pmc_mfma.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
dur_by_id = {}
for f in glob.glob("$R/gpurun_out/pmc_mfma/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur_by_id[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
dur = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_ldl_" not in k: continue
        k = k.split("(anonymous namespace)::")[-1].split("(")[0][:48]
        big = ""
        if k.startswith("k_ldl_update<") and dur_by_id.get(r["Dispatch_Id"], 0) > 300: big = " [launches > 300 us]"
        k += big
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"]); dur[k] += dur_by_id.get(r["Dispatch_Id"], 0.0)
with open("$R/gpurun_out/pmc_mfma_bench.txt", "w") as o:
    o.write("# per kernel, summed over its launches: MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x time x clock); clock = GRBM_GUI_ACTIVE / 8 / time\n")
    for k in sorted(tot, key=lambda k: -dur[k]):
        c = tot[k]; us = dur[k]
        if us <= 0 or "GRBM_GUI_ACTIVE" not in c: continue
        clock = c["GRBM_GUI_ACTIVE"] / 8 / us / 1e3
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * us * 1e3 * clock)
        line = f"{k:70s} launches {len(n[k]):5d} time {us/1e3:9.2f} ms  clock {clock:.3f} GHz  MFMA pipes busy {util*100:5.1f} %  => {util*clock/2.4*100:5.1f} % of the 2.4 GHz peak"
        print(line); o.write(line + "\n")
PY
find $R/gpurun_out/pmc_mfma -name "*.csv" -size +2M -delete
