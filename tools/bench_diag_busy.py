"""The diagonal-tile kernel alone on the device and beside pure-MFMA filler work on a second stream: per-launch us.
usage: bench_diag_busy.py [n] [fill_blocks] [fill_iters]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib
ba, L = _benchlib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L.ba_debug_diag_busy.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
for blocks, iters in ((int(sys.argv[2]), int(sys.argv[3])),) if len(sys.argv) > 3 else ((400, 40000), (64, 40000), (8, 40000), (1600, 10000)):
    out = np.zeros(2 * n)
    assert L.ba_debug_diag_busy(n, blocks, iters, out.ctypes.data) == 0
    print(f"filler {blocks} workgroups x {iters} iterations")
    print("  alone :", " ".join(f"{v:5.1f}" for v in out[:n]))
    print("  beside:", " ".join(f"{v:5.1f}" for v in out[n:]))
