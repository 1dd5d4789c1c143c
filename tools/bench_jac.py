"""jac_coord micro-benchmark on the Venice shape: 0 real, 1 no arithmetic, 2 no stores, 3 neither."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib
import torch
ba, L = _benchlib.load()
name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 0]
prob = ba.synthetic.make_named(name)
nlp = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
x = torch.from_numpy(prob["x0"]).cuda(); vals = torch.empty(24 * prob["nobs"], dtype=torch.float64, device="cuda")
L.ba_debug_jac_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
nbytes = (208.0 + 8.0 * nlp.meta.nvar / prob["nobs"]) * prob["nobs"]
for v in variants:
    ms = C.c_double(0)
    rc = L.ba_debug_jac_bench(nlp.handle, C.c_void_p(x.data_ptr()), C.c_void_p(vals.data_ptr()), v, 10, C.byref(ms))
    print(f"variant {v}: {ms.value:.4f} ms  {24*prob['nobs']/ms.value/1e3:.0f} Mnnz/s  {nbytes/ms.value/1e6:.0f} GB/s algorithmic (rc {rc})", flush=True)
