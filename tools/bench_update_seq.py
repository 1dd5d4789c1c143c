"""Sequences of trailing-update launches (nonzero operands) with per-launch times: how the f64 matrix rate behaves after a
cold start, after a ramp of small launches, and along the launch sizes of a real factorisation.
usage: bench_update_seq.py [nt]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib
ba, L = _benchlib.load()
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 126
L.ba_debug_update_seq.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def run(tag, ms_list, gaps=None):
    m = np.asarray(ms_list, dtype=np.int32)
    g = np.zeros(len(m), dtype=np.int32) if gaps is None else np.asarray(gaps, dtype=np.int32)
    out = np.zeros(len(m))
    rc = L.ba_debug_update_seq(nt, len(m), m.ctypes.data, g.ctypes.data, out.ctypes.data)
    assert rc == 0
    tiles = m.astype(np.int64) * (m + 1) // 2
    print(f"--- {tag}: total {out.sum():.2f} ms, {tiles.sum() * 2 * 128 * 128 * 256 / out.sum() / 1e9:.1f} TFLOP/s")
    print("  m      : " + " ".join(f"{v:6d}" for v in m[:40]))
    print("  us/tile: " + " ".join(f"{1e3 * t / n:6.3f}" for t, n in zip(out[:40], tiles[:40])))
    return out


big = nt - 2


def show(tag, ms_list, gaps=None):
    m = np.asarray(ms_list, dtype=np.int32)
    out = run_raw(m, gaps)
    tiles = np.where(m > 0, m.astype(np.int64) * (m + 1) // 2, 0)
    ok = m > 0
    print(f"--- {tag}: updates {out[ok].sum():.2f} ms ({tiles.sum() * 2 * 128 * 128 * 256 / out[ok].sum() / 1e9:.1f} TFLOP/s), everything {out.sum():.2f} ms")
    print("  m      : " + " ".join(f"{v:6d}" for v in m[:44]))
    print("  us/tile: " + " ".join((f"{1e3 * t / n:6.3f}" if n else f"{t:5.1f}m") for t, n in zip(out[:44], tiles[:44])))


def run_raw(m, gaps):
    g = np.zeros(len(m), dtype=np.int32) if gaps is None else np.asarray(gaps, dtype=np.int32)
    out = np.zeros(len(m))
    assert L.ba_debug_update_seq(nt, len(m), m.ctypes.data, g.ctypes.data, out.ctypes.data) == 0
    return out


ramp = list(range(8, 65, 8))
show("20 ms idle, 16 big", [big] * 16, [20000] + [0] * 15)
show("20 ms idle, ramp 8..64, 16 big", ramp + [big] * 16, [20000] + [0] * 23)
show("20 ms idle, 16 big (again)", [big] * 16, [20000] + [0] * 15)
show("20 ms idle, ramp x3, 16 big", [v for v in ramp for _ in range(3)] + [big] * 16, [20000] + [0] * 39)
show("20 ms idle, 40 fills (memory-bound, ~8 ms), 16 big", [-40] + [big] * 16, [20000] + [0] * 16)
show("20 ms idle, 40 fills, ramp, 16 big", [-40] + ramp + [big] * 16, [20000] + [0] * 24)
show("20 ms idle, 16 big (third)", [big] * 16, [20000] + [0] * 15)
show("20 ms idle, 4 x m=64, 16 big", [64] * 4 + [big] * 16, [20000] + [0] * 19)
show("20 ms idle, 1 x m=32, 16 big", [32] + [big] * 16, [20000] + [0] * 16)

# the LM loop in miniature: [memory-bound phase] + the 62 updates of one factorisation, five times in one stream
fact = list(range(big, 1, -2))
for fills in (35, 10, 0):
    seq = ([-fills] if fills else []) + fact
    m = np.asarray(seq * 5, dtype=np.int32)
    out = run_raw(m, None)
    per = out.reshape(5, len(seq))
    upd = per[:, 1:] if fills else per
    print(f"--- 5 x [{fills} fills + factorisation order]: update sums per factorisation (ms): " + " ".join(f"{v:.2f}" for v in upd.sum(1))
          + (f"; fills {per[:, 0].mean():.2f} ms" if fills else ""))
    t = np.asarray(fact[:8]); tl = t * (t + 1) // 2
    print("    us/tile of the first 8 launches, last repetition: " + " ".join(f"{1e3 * v / n:.3f}" for v, n in zip(upd[-1, :8], tl)))
