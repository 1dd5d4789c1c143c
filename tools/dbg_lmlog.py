"""Side-by-side LM logs, device vs oracle, for the hard-start cases of tests/test_gpu_parity.py (debugging aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ba = ge.load_package(); orc = ge.load_oracle()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _hard_start, _TIGHT

def show(p, x0, facto, ls, kw, graph=None):
    if graph is not None:
        os.environ["BA_LM_GRAPH"] = graph
    m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
    st = ba.Levenberg_Marquardt(ba.FeasibilityResidual(m), facto, "AMD", "None", ls, x=x0, **kw)
    rc, x_ref, st_ref, log_ref = orc.lm_solve(p["ncams"], p["npnts"], p["cam_idx1"], p["pnt_idx1"], p["pt2d"], x0, variant=1, linesearch=ls, facto=facto, **kw)
    print(f"--- facto {facto} linesearch {ls} {kw.get('nu_d')} {kw.get('delta_d')} graph {graph}: device {st.iter} {st.status} oracle {st_ref.iter}")
    for k in range(min(len(st.log), len(log_ref), 14)):
        a, b = st.log[k], log_ref[k]
        print("  dev %3d f %.10e df %.3e g %.6e lam %.8e nd %.8e rho %+.6e %d" % (a[:7] + (int(a[7]),)))
        print("  orc %3d f %.10e df %.3e g %.6e lam %.8e nd %.8e rho %+.6e %d" % tuple(b))
    m.close()

p = ba.synthetic.make_problem(12, 400, 1800, seed=11)
x0 = _hard_start(p)
show(p, x0, "LDL", False, dict(nu_d=9.0, delta_d=2.0, ite_max=40, **_TIGHT), graph="1")
show(p, x0, "LDL", False, dict(nu_d=9.0, delta_d=2.0, ite_max=40, **_TIGHT), graph="0")
show(p, x0, "LDL", True, dict(nu_d=27.0, delta_d=2.0, ite_max=40, **_TIGHT), graph="0")
p2 = ba.synthetic.make_problem(6, 80, 320, seed=9)
show(p2, _hard_start(p2, 1.0, 0.3, 5), "QR", True, dict(nu_d=27.0, delta_d=3.0, ite_max=25, **_TIGHT), graph="0")
