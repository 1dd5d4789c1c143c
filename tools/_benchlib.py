"""Load the package over tools/libba_bench.so: the product sources plus the probe / ablation entry points of
csrc/bench/ (`make -C bundleadjustment.jl_amd/csrc bench` builds it).  The product library has none of them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def load():
    path = os.path.join(ROOT, "tools", os.environ.get("BA_BENCH_LIB", "libba_bench.so"))
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bundleadjustment.jl_amd", "csrc"), "bench"])
    ba = ge.load_package()
    assert ba._lib._lib is None, "the product library is already loaded in this process"
    ba._lib.LIB_PATH = path
    return ba, ba._lib.lib()
