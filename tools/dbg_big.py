import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ba = ge.load_package(); orc = ge.load_oracle()
ncams, npnts, nobs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = ba.synthetic.make_problem(ncams, npnts, nobs, seed=3)
m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
r = m.cons(p["x0"]); r_ref = orc.residuals(p["cam_idx1"], p["pnt_idx1"], p["x0"], p["pt2d"], p["npnts"])
dr = np.abs(r - r_ref); print("residual max diff", dr.max(), "at", dr.argmax(), "n bad", int((dr > 1e-9).sum()))
v = m.jac_coord(p["x0"]); v_ref = orc.jac_coord(p["cam_idx1"], p["pnt_idx1"], p["x0"], p["npnts"])
dv = np.abs(v - v_ref).reshape(-1, 24).max(1) / np.abs(v_ref).reshape(-1, 24).max(1)
bad = np.flatnonzero(dv > 1e-10); print("jac max rel diff", dv.max(), "n bad obs", bad.size, "first bad", bad[:10], "last bad", bad[-5:])
rows, cols = orc.jac_structure(p["cam_idx1"], p["pnt_idx1"], p["npnts"])
g = orc.mul_sparse(cols, rows, v_ref, r_ref, m.meta.nvar)
j = m.jtprod_coo(v_ref, r_ref)
dj = np.abs(j - g); print("jtr (oracle J, r) max diff rel", dj.max() / np.abs(g).max(), "argmax", dj.argmax(), "of nvar", m.meta.nvar, "3*npnts", 3 * npnts)
cams = p["cam_idx1"][bad] - 1
uc, cnt = np.unique(cams, return_counts=True)
print("bad cameras", uc[:20], "counts", cnt[:20], "n", uc.size)
X = p["x0"]; C = X[3 * npnts:].reshape(-1, 9)
print("theta of bad cams", np.linalg.norm(C[uc[:10], :3], axis=1))
print("theta range all", np.linalg.norm(C[:, :3], axis=1).min(), np.linalg.norm(C[:, :3], axis=1).max())
k = bad[0]
print("obs", k, "cam", cams[0], "gpu", v.reshape(-1, 24)[k], "\nref", v_ref.reshape(-1, 24)[k])
tot = np.bincount(p["cam_idx1"] - 1, minlength=ncams)
print("bad fraction within those cams", cnt[:10] / tot[uc[:10]])
np.set_printoptions(precision=17, linewidth=200)
print("diff", (v.reshape(-1, 24)[k] - v_ref.reshape(-1, 24)[k]))
pi = p["pnt_idx1"][k] - 1
print("X", X[3*pi:3*pi+3], "C", C[cams[0]])
# neighbours in the same wave batch: which lanes are bad?
print("bad lanes mod 64 histogram", np.bincount(bad % 64, minlength=64))
print("bad batch-in-wave (obs//64 % 4) histogram", np.bincount((bad // 64) % 4, minlength=4))
