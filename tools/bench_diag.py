import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib
ba, L = _benchlib.load()
cyc = (C.c_double * 118)(); ms = C.c_double(0)
L.ba_debug_diag_stamps.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
rc = L.ba_debug_diag_stamps(cyc, C.byref(ms))
names = ["load", "P phases x8 (wave 0: diagonal-block update + 16 pivots; workers: tasks)", "A phases x8 (row solves, 16x16 inverse, deferred tasks)",
         "block-column update phase (four-wave form only)", "tail (last block row of the inverse, D)", "store phase (gone)"]
tot = sum(cyc[:6])
print(f"rc {rc}: kernel {ms.value*1e3:.1f} us (unstamped run); stamped run, s_memtime cycles per phase (barrier to barrier):")
for n, c in zip(names, cyc): print(f"  {c:10.0f}  {100*c/tot:5.1f}%  {n}")
print("busy cycles of wave w (columns) while wave 0 factors diagonal block s + 1 (rows):")
for st in range(7): print("  s=%d " % st + " ".join("%6.0f" % cyc[6 + 8 * st + w] for w in range(8)))
print("busy cycles of wave w in the row-solve phase A(s + 1) (waves 0-1: row solves, 3: inverse of the 16 x 16 block, others: deferred tasks):")
for st in range(7): print("  s=%d " % st + " ".join("%6.0f" % cyc[62 + 8 * st + w] for w in range(8)))
