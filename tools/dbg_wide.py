"""Debug: LM step on a very wide camera system (many cameras, few points): checks the normal-equation residual."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ba = ge.load_package(); orc = ge.load_oracle()
ncams, npnts, nobs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lam = float(sys.argv[4]) if len(sys.argv) > 4 else 50.0
p = ba.synthetic.make_problem(ncams, npnts, nobs, seed=3)
m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
t = time.time()
try:
    d, half, jtr = ba.lm_step(m, p["x0"], lam)
except Exception as e:
    print("FAILED:", e); sys.exit(0)
print("step time", time.time() - t)
rows, cols = orc.jac_structure(p["cam_idx1"], p["pnt_idx1"], p["npnts"])
vals = orc.jac_coord(p["cam_idx1"], p["pnt_idx1"], p["x0"], p["npnts"])
r = orc.residuals(p["cam_idx1"], p["pnt_idx1"], p["x0"], p["pt2d"], p["npnts"])
nvar = m.meta.nvar
Jd = orc.mul_sparse(rows, cols, vals, d, 2 * p["nobs"])
g = orc.mul_sparse(cols, rows, vals, r, nvar)
res = orc.mul_sparse(cols, rows, vals, Jd, nvar) + lam * d + g
print("n =", 9 * ncams, "|res|/|g| =", np.linalg.norm(res) / np.linalg.norm(g), " |jtr-g|/|g| =", np.linalg.norm(jtr - g) / np.linalg.norm(g),
      " half:", half, 0.5 * np.sum((Jd + r) ** 2), " nan in d:", int(np.isnan(d).sum()))
