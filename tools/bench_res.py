"""Residual / Jacobian launch time on the Venice shape: `reps` back-to-back launches on one stream, events on that stream.
BA_CAM_FUSED=0 gives the separate k_cam_pre launch.  usage: python tools/bench_res.py [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
ba = ge.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = ba.synthetic.make_named("venice-1778")
nlp = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
L = ba._lib.lib()
stream = torch.cuda.Stream()
sp = C.c_void_p(stream.cuda_stream)
x = torch.from_numpy(p["x0"]).cuda()
r = torch.empty(2 * p["nobs"], dtype=torch.float64, device="cuda")
v = torch.empty(24 * p["nobs"], dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0.record(stream)
        for _ in range(reps): fn()
        e1.record(stream)
        torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return out
res = timeit(lambda: ba._lib.check(L.ba_residual_dev(nlp.handle, C.c_void_p(x.data_ptr()), C.c_void_p(r.data_ptr()), sp)))
jac = timeit(lambda: ba._lib.check(L.ba_jac_coord_dev(nlp.handle, C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()), sp)))
nb = (48 + 8 * (3 * p["npnts"] + 9 * p["ncams"]) / p["nobs"]) * p["nobs"]
print(f"fused={os.environ.get('BA_CAM_FUSED', '1')} residual us {[round(t, 1) for t in res]} -> {nb / (min(res) * 1e-6) / 8e12:.3f} of HBM; "
      f"jacobian us {[round(t, 1) for t in jac]}")
