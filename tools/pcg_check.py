import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ba = ge.load_package()
for name in ("ladybug-49", "dubrovnik-356"):
    prob = ba.synthetic.make_named(name)
    m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
    fr = ba.FeasibilityResidual(m)
    for facto, kw in (("LDL", {}), ("PCG", {}), ("PCG", {"pcg_tol": 1e-3}), ("PCG", {"pcg_tol": 1e-1})):
        t = time.time()
        st = ba.Levenberg_Marquardt(fr, facto, "AMD", "None", False, log=False, **kw)
        print(name, facto, kw, "iter", st.iter, "acc/rej", st.n_accepted, st.n_rejected, "obj %.12e" % st.objective, st.status, "n_cg", st.n_cg, "loop %.3fs" % st.loop_time, flush=True)
    m.close()
