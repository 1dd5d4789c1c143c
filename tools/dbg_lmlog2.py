import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ba = ge.load_package(); orc = ge.load_oracle()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _hard_start, _TIGHT
p = ba.synthetic.make_problem(12, 400, 1800, seed=11)
x0 = _hard_start(p)
kw = dict(nu_d=9.0, delta_d=3.0, ite_max=40, **_TIGHT)
m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
st = ba.Levenberg_Marquardt(ba.FeasibilityResidual(m), "LDL", "AMD", "None", True, x=x0, **kw)
rc, x_ref, st_ref, log_ref = orc.lm_solve(p["ncams"], p["npnts"], p["cam_idx1"], p["pnt_idx1"], p["pt2d"], x0, variant=1, linesearch=True, **kw)
for k in range(min(len(st.log), len(log_ref), 6)):
    a, b = st.log[k], log_ref[k]
    print("  dev %3d f %.10e df %.3e g %.6e lam %.8e nd %.8e rho %+.6e %d" % (a[:7] + (int(a[7]),)))
    print("  orc %3d f %.10e df %.3e g %.6e lam %.8e nd %.8e rho %+.6e %d" % tuple(b))
