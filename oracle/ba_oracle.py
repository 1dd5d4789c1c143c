"""ctypes view of oracle/libba_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
See oracle/ba_oracle.c for the reference file:line each entry follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# libgomp's default active spinning starves the (single-threaded) factorisation under a CPU quota: idle threads sleep.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

STATUS = ["small_step", "first_order", "small_residual", "acceptable", "neg_pred", "exception", "max_iter"]


class LMOpts(C.Structure):
    _fields_ = [("variant", C.c_int), ("normalize", C.c_int), ("linesearch", C.c_int), ("facto_f32", C.c_int),
                ("ite_max", C.c_int),
                ("restol", C.c_double), ("satol", C.c_double), ("srtol", C.c_double), ("oatol", C.c_double),
                ("ortol", C.c_double), ("atol", C.c_double), ("rtol", C.c_double),
                ("nu_d", C.c_double), ("nu_m", C.c_double), ("lam", C.c_double), ("delta_d", C.c_double),
                ("facto_time_cap_s", C.c_double), ("max_iter_timed", C.c_int), ("facto_qr", C.c_int)]


class LMStats(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int),
                ("objective", C.c_double), ("dual_feas", C.c_double), ("elapsed_s", C.c_double),
                ("lambda_final", C.c_double),
                ("t_residual", C.c_double), ("t_jac", C.c_double), ("t_assemble", C.c_double),
                ("t_facto", C.c_double), ("t_solve", C.c_double), ("t_jtr", C.c_double),
                ("n_facto", C.c_int), ("n_jac", C.c_int), ("n_res", C.c_int),
                ("lnz", C.c_int64), ("facto_work_total", C.c_double), ("facto_work_done", C.c_double),
                ("n_log", C.c_int), ("n_sparse", C.c_int), ("t_analyse", C.c_double),
                ("t_facto_head", C.c_double), ("facto_work_head", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libba_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_num_threads.restype = C.c_int
    L.orc_projection.argtypes = [f64p, f64p, f64p]
    L.orc_P1.argtypes = [f64p, f64p, f64p, f64p]
    L.orc_projection_f32.argtypes = [f32p, f32p, f32p]
    L.orc_residuals.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, f64p]
    L.orc_residuals_f32.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f32p, f32p, f32p]
    L.orc_jac_structure.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p]
    L.orc_jac_coord.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f64p, f64p]
    L.orc_jac_coord_f32.argtypes = [C.c_int64, C.c_int64, i64p, i64p, f32p, f32p]
    L.orc_mul_sparse.argtypes = [C.c_int64, i64p, i64p, f64p, f64p, C.c_int64, f64p]
    L.orc_sparse.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f64p, i64p, i64p, f64p]
    L.orc_normalize_qr_a.argtypes = [i64p, f64p, f64p, C.c_int64]
    L.orc_normalize_qr_j.argtypes = [i64p, f64p, f64p, C.c_int64]
    L.orc_denormalize_qr.argtypes = [i64p, f64p, f64p, C.c_int64]
    L.orc_normalize_ldl.argtypes = [i64p, f64p, f64p, C.c_int64, C.c_int64]
    L.orc_denormalize_ldl.argtypes = [i64p, f64p, f64p, C.c_int64, C.c_int64]
    L.orc_denormalize_vect.argtypes = [f64p, f64p, C.c_int64]
    L.orc_ldl_solve.argtypes = [C.c_int64, i64p, i64p, f64p, i64p, f64p, C.POINTER(C.c_int64), C.c_void_p]
    L.orc_ldl_solve.restype = C.c_int
    L.orc_lm_solve.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, C.POINTER(LMOpts),
                               C.POINTER(LMStats), C.c_void_p, C.c_int]
    L.orc_lm_solve.restype = C.c_int
    L.orc_lm_solve_f32.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f32p, f32p, C.POINTER(LMOpts),
                                   C.POINTER(LMStats), C.c_void_p, C.c_int]
    L.orc_lm_solve_f32.restype = C.c_int
    L.orc_lm_step.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, C.c_double, f64p,
                              C.c_void_p, C.c_void_p]
    L.orc_lm_step.restype = C.c_int
    L.orc_lm_step_perm.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, C.c_double, C.c_void_p, f64p,
                                   C.c_void_p, C.c_void_p]
    L.orc_lm_step_perm.restype = C.c_int
    L.orc_qr_lstsq.argtypes = [C.c_int64, C.c_int64, f64p, f64p]
    L.orc_qr_lstsq.restype = C.c_int
    _LIB = L
    return L


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def projection(X, Cam):
    out = np.zeros(2)
    lib().orc_projection(_f64(X), _f64(Cam), out)
    return out


def P1(r, t, X):
    out = np.zeros(3)
    lib().orc_P1(_f64(r), _f64(t), _f64(X), out)
    return out


def residuals(cam_idx1, pnt_idx1, x, pt2d, npnts):
    """cons!(nlp, x, cx): projection residuals minus pt2d (1-based indices, x = [points; cameras])."""
    nobs = len(cam_idx1)
    x = np.asarray(x)
    if x.dtype == np.float32:
        cx = np.zeros(2 * nobs, dtype=np.float32)
        lib().orc_residuals_f32(nobs, npnts, _i64(cam_idx1), _i64(pnt_idx1), np.ascontiguousarray(x),
                                np.ascontiguousarray(pt2d, dtype=np.float32), cx)
        return cx
    cx = np.zeros(2 * nobs)
    lib().orc_residuals(nobs, npnts, _i64(cam_idx1), _i64(pnt_idx1), _f64(x), _f64(pt2d), cx)
    return cx


def jac_structure(cam_idx1, pnt_idx1, npnts):
    nobs = len(cam_idx1)
    rows = np.zeros(24 * nobs, dtype=np.int64)
    cols = np.zeros(24 * nobs, dtype=np.int64)
    lib().orc_jac_structure(nobs, npnts, _i64(cam_idx1), _i64(pnt_idx1), rows, cols)
    return rows, cols


def jac_coord(cam_idx1, pnt_idx1, x, npnts):
    nobs = len(cam_idx1)
    x = np.asarray(x)
    if x.dtype == np.float32:
        vals = np.zeros(24 * nobs, dtype=np.float32)
        lib().orc_jac_coord_f32(nobs, npnts, _i64(cam_idx1), _i64(pnt_idx1), np.ascontiguousarray(x), vals)
        return vals
    vals = np.zeros(24 * nobs)
    lib().orc_jac_coord(nobs, npnts, _i64(cam_idx1), _i64(pnt_idx1), _f64(x), vals)
    return vals


def mul_sparse(rows1, cols1, vals, x, nout):
    xr = np.zeros(nout)
    lib().orc_mul_sparse(nout, _i64(rows1), _i64(cols1), _f64(vals), _f64(x), len(vals), xr)
    return xr


def sparse(I1, J1, V, m, n):
    nnz = len(V)
    colptr = np.zeros(n + 1, dtype=np.int64)
    rowval = np.zeros(nnz, dtype=np.int64)
    nzval = np.zeros(nnz)
    lib().orc_sparse(m, n, nnz, _i64(I1), _i64(J1), _f64(V), colptr, rowval, nzval)
    return colptr, rowval, nzval


def ldl_solve(colptr, rowval, nzval, P, b):
    n = len(colptr) - 1
    x = _f64(b).copy()
    lnz = C.c_int64(0)
    D = np.zeros(n)
    rc = lib().orc_ldl_solve(n, _i64(colptr), _i64(rowval), _f64(nzval), _i64(P), x, C.byref(lnz),
                             D.ctypes.data_as(C.c_void_p))
    return rc, x, lnz.value, D


def qr_lstsq(A, b):
    """argmin |A x - b| by the dense Householder QR that stands in for SPQR (myqr + solve_qr!, src/qr_aux.jl:13-55)."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    Af = np.ascontiguousarray(A.T).copy()  # column-major m x n
    bb = _f64(b).copy()
    rc = lib().orc_qr_lstsq(m, n, Af.reshape(-1), bb)
    return rc, bb[:n]


def lm_step(ncams, npnts, cam_idx1, pnt_idx1, pt2d, x, lam, cam_perm1=None):
    """cam_perm1: the cameras' elimination order handed to ldl_analyse (src/ldl_aux.jl:246-283); None: the identity."""
    nobs = len(cam_idx1)
    nvar = 9 * ncams + 3 * npnts
    delta = np.zeros(nvar)
    dr = np.zeros(2 * nobs)
    jtr = np.zeros(nvar)
    perm = None if cam_perm1 is None else _i64(cam_perm1)
    rc = lib().orc_lm_step_perm(ncams, npnts, nobs, _i64(cam_idx1), _i64(pnt_idx1), _f64(pt2d), _f64(x), float(lam),
                                perm.ctypes.data_as(C.c_void_p) if perm is not None else None,
                                delta, dr.ctypes.data_as(C.c_void_p), jtr.ctypes.data_as(C.c_void_p))
    return rc, delta, dr, jtr


def lm_solve(ncams, npnts, cam_idx1, pnt_idx1, pt2d, x0, variant=1, normalize=0, linesearch=False, facto_f32=False,
             ite_max=-1, lam=-1.0, facto_time_cap_s=0.0, max_iter_timed=0, log_cap=512, facto="LDL", **tols):
    """Levenberg_Marquardt(...) of lm.jl (variant=1) or LevenbergMarquardt.jl (variant=0); facto="QR": the :QR branch of
    lm.jl with a dense Householder QR (small problems only: the matrix is densified)."""
    nobs = len(cam_idx1)
    # facto_f32: False/0 Float64, True/1 Float32, 2 Float16 (emulated: Float32 operation + rounding to binary16, as Julia)
    o = LMOpts(variant=variant, normalize=normalize, linesearch=int(linesearch), facto_f32=int(facto_f32),
               ite_max=ite_max, restol=tols.get("restol", -1.0), satol=tols.get("satol", -1.0),
               srtol=tols.get("srtol", -1.0), oatol=tols.get("oatol", -1.0), ortol=tols.get("ortol", -1.0),
               atol=tols.get("atol", -1.0), rtol=tols.get("rtol", -1.0), nu_d=tols.get("nu_d", -1.0),
               nu_m=tols.get("nu_m", -1.0), lam=lam, delta_d=tols.get("delta_d", -1.0), facto_time_cap_s=facto_time_cap_s, max_iter_timed=max_iter_timed,
               facto_qr=int(facto == "QR"))
    st = LMStats()
    x = _f64(x0).copy()
    log = np.zeros((log_cap, 8))
    rc = lib().orc_lm_solve(ncams, npnts, nobs, _i64(cam_idx1), _i64(pnt_idx1), _f64(pt2d), x, C.byref(o),
                            C.byref(st), log.ctypes.data_as(C.c_void_p), log_cap)
    return rc, x, st, log[: st.n_log]


def lm_solve_f32(ncams, npnts, cam_idx1, pnt_idx1, pt2d, x0, variant=1, normalize=0, linesearch=False, ite_max=-1, lam=-1.0,
                 max_iter_timed=0, log_cap=512, **tols):
    """Levenberg_Marquardt on a Float32 model (eltype(x) = Float32, facto_type = Float32 = its default): every scalar with
    the width Julia's promotion rules give it (oracle/ba_oracle.c, orc_lm_solve_f32).  Tolerances / parameters passed here
    are Float64 values, as literals are in the reference's own call (src/diffprecsions.jl:22); omitted ones are the
    eps(Float32)-derived Float32 defaults."""
    nobs = len(cam_idx1)
    o = LMOpts(variant=variant, normalize=normalize, linesearch=int(linesearch), facto_f32=1,
               ite_max=ite_max, restol=tols.get("restol", -1.0), satol=tols.get("satol", -1.0),
               srtol=tols.get("srtol", -1.0), oatol=tols.get("oatol", -1.0), ortol=tols.get("ortol", -1.0),
               atol=tols.get("atol", -1.0), rtol=tols.get("rtol", -1.0), nu_d=tols.get("nu_d", -1.0),
               nu_m=tols.get("nu_m", -1.0), lam=lam, delta_d=tols.get("delta_d", -1.0), facto_time_cap_s=0.0,
               max_iter_timed=max_iter_timed, facto_qr=0)
    st = LMStats()
    x = np.ascontiguousarray(x0, dtype=np.float32).copy()
    log = np.zeros((log_cap, 8))
    rc = lib().orc_lm_solve_f32(ncams, npnts, nobs, _i64(cam_idx1), _i64(pnt_idx1), np.ascontiguousarray(pt2d, dtype=np.float32),
                                x, C.byref(o), C.byref(st), log.ctypes.data_as(C.c_void_p), log_cap)
    return rc, x, st, log[: st.n_log]
