"""f64 MFMA ceiling probe (ba_debug_mfma_probe): 0 = MFMAs only, 1 = with the update kernel's LDS operand reads, 2 = 4x8
blocks per wave at one wave per SIMD."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ba = ge.load_package()
L = ba._lib.lib()
L.ba_debug_mfma_probe.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
for mode in (0, 1, 2, 0, 1, 2):
    tf = C.c_double(0)
    rc = L.ba_debug_mfma_probe(mode, 2000, C.byref(tf))
    print(f"mode {mode}: {tf.value:.1f} TFLOP/s (rc {rc})", flush=True)
