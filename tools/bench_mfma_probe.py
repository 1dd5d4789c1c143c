"""f64 MFMA ceiling probes (ba_debug_mfma_probe): 0 = MFMAs only, 1 = with the update kernel's LDS operand reads, 2 = 4x8
blocks per wave at one wave per SIMD (no staging); 3 = full wave-private staging loop, shipped geometry (64x64 per wave, two
workgroups per CU), 4 = the same loop with 64x128 per wave, one workgroup per CU."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib
ba, L = _benchlib.load()
L.ba_debug_mfma_probe.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
for mode in ([int(m) for m in sys.argv[1].split(',')] if len(sys.argv) > 1 else (0, 1, 2, 3, 4, 5, 3, 5)):
    tf = C.c_double(0)
    rc = L.ba_debug_mfma_probe(mode, 2000 if mode <= 2 else int(sys.argv[2]) if len(sys.argv) > 2 else 16, C.byref(tf))
    print(f"mode {mode}: {tf.value:.1f} TFLOP/s (rc {rc})", flush=True)
