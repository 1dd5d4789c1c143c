"""Micro-benchmark of the bulk trailing update kernel (K = 256 pair update) on an nt x nt tile matrix.
variants: 0 real (wave-private LDS staging), 1 store-only epilogue, 2 L2-resident operands, 8 workgroup-shared staging
with barriers (first version), 9 = 8 + store-only, 16 LDS-DMA staging, 32 K = 512, 64 accumulators started from -C,
128 direct operand loads without LDS (+256 half the loads, +512 L1-resident operands, +1024 / +2048 / +4096 a workgroup
barrier every 1 / 4 / 2 K groups), 8192 super-block enumeration of the triangle; +1 on any of them: store-only epilogue.
BA_BENCH_NONZERO=1 BA_BENCH_ENTROPY=1: full-entropy mantissas (what a real matrix holds; the chip clocks lower on them)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib
ba, L = _benchlib.load()
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 126
L.ba_debug_update_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for v in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 1, 2, 8, 9, 16, 17, 0, 16)):
    m = nt - (4 if v & 32 else 2)  # variants 32 / 33: K = 512 (four panels per pass; 33: store-only epilogue)
    flops = m * (m + 1) / 2 * 2 * 128 * 128 * (512 if v & 32 else 256)
    ms = C.c_double(0)
    rc = L.ba_debug_update_bench(nt, v, int(os.environ.get('BA_BENCH_REPS', '5')), C.byref(ms))
    print(f"variant {v}: {ms.value:.3f} ms  {flops / ms.value / 1e9:.1f} TFLOP/s (rc {rc})", flush=True)
