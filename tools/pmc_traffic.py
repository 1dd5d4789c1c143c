"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command -> JSON.

Units and corrections for gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, HBM counters): one counter unit is 1024 B
(checked here against a kernel of known store volume: k_jac_coord writes 24*8*nobs B = 960.37 MB at nobs = 5001946
and WRITE_SIZE reads 9.379e5); FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled; WRITE_SIZE is
exact."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402  (the stamp bench.py checks before it reports `traffic`)

UNIT = 1024.0  # bytes per counter unit (checked against k_jac_coord's known store volume, see above)


def per_kernel(d, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            k = k.split("(anonymous namespace)::")[-1].split("(")[0].strip()
            tot[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(disp[k])) for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    rd = 2.0 * f * UNIT / max(nf, 1)
    wr = w * UNIT / max(nw, 1)
    out[k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
commit = os.environ.get("BA_COMMIT")  # the GPU box has no .git: tools/collect_pmc.sh is given the commit by its caller
if not commit:
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except (OSError, subprocess.CalledProcessError):
        commit = None
json.dump({"kernel_sources_sha256": kernel_sources_sha(), "commit": commit,
           "note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE exact, 1024 B per unit; averages over all launches "
                   "of the kernel in the profiled command", "kernels": out}, sys.stdout, indent=1)
print()
