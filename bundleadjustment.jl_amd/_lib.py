"""ctypes binding of libba_hip.so (include/ba_hip.h).  There is no CPU fallback: if the HIP library is
missing or no MI355X is visible, every compute entry raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libba_hip.so")

BA_OK = 0
ERR_NAMES = {1: "BA_ERR_ARG", 2: "BA_ERR_HIP", 3: "BA_ERR_IO", 4: "BA_ERR_ZERO_PIVOT", 5: "BA_ERR_NAN_STEP",
             6: "BA_ERR_COMM"}
STATUS = {-1: "unknown", 0: "small_step", 1: "first_order", 2: "small_residual", 3: "acceptable", 4: "neg_pred",
          5: "exception", 6: "max_iter"}


class BAError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class SQDException(BAError):
    """Zero pivot in the LDL' factorisation (reference: src/ldl_aux.jl:45-47,199)."""


class LMOpts(C.Structure):
    _fields_ = [("variant", C.c_int), ("facto", C.c_int), ("normalize", C.c_int), ("linesearch", C.c_int),
                ("facto_type", C.c_int), ("ite_max", C.c_int), ("verbose", C.c_int), ("x_f32", C.c_int),
                ("restol", C.c_double), ("satol", C.c_double), ("srtol", C.c_double), ("oatol", C.c_double),
                ("ortol", C.c_double), ("atol", C.c_double), ("rtol", C.c_double),
                ("nu_d", C.c_double), ("nu_m", C.c_double), ("lam", C.c_double), ("delta_d", C.c_double),
                ("max_time", C.c_double), ("pcg_tol", C.c_double), ("pcg_max_iter", C.c_int), ("perm", C.c_int)]


class LMStats(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("n_accepted", C.c_int), ("n_rejected", C.c_int),
                ("n_residual", C.c_int), ("n_jacobian", C.c_int), ("n_factor", C.c_int), ("n_cg", C.c_int),
                ("objective", C.c_double), ("dual_feas", C.c_double), ("lambda_final", C.c_double),
                ("elapsed_s", C.c_double), ("loop_s", C.c_double)]


LOG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                     C.c_double, C.c_int)
COMM_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)
COMM_ALLREDUCE_F64, COMM_REDUCE_F64, COMM_BCAST_BYTES, COMM_REDUCE_F32, COMM_REDUCE_SCATTER_F64, COMM_REDUCE_SCATTER_F32 = 0, 1, 2, 3, 4, 5
COMM_OPS = 6
COMM_OP_NAMES = ("allreduce_f64", "reduce_f64", "bcast_bytes", "reduce_f32", "reduce_scatter_f64", "reduce_scatter_f32")
COMM_ID_BYTES = 128

# every symbol include/ba_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "ba_last_error", "ba_device_count", "ba_device_info", "ba_read_bal_header", "ba_read_bal", "ba_read_bal_f32",
    "ba_problem_create", "ba_problem_destroy", "ba_problem_dims", "ba_residual", "ba_residual_f32",
    "ba_jac_structure", "ba_jac_coord", "ba_jac_coord_f32", "ba_jtr", "ba_residual_dev", "ba_residual_f32_dev",
    "ba_jac_structure_dev", "ba_jac_coord_dev", "ba_jac_coord_f32_dev", "ba_jtr_dev", "ba_dev_malloc", "ba_dev_free",
    "ba_memcpy_h2d", "ba_memcpy_d2h", "ba_memcpy_h2d_on", "ba_memcpy_d2h_on", "ba_synchronize", "ba_lm_solve", "ba_lm_solve_dev", "ba_comm_get_unique_id", "ba_lm_set_comm_rccl",
    "ba_lm_set_comm_hook", "ba_comm_stats", "ba_comm_stats_ops", "ba_dist_layout",
    "ba_lm_step", "ba_lm_step_f32", "ba_lm_step_pcg", "ba_lm_schur_pattern", "ba_lm_schur_memory", "ba_schur_ordering", "ba_lm_set_ordering", "ba_lm_schur_ordering", "ba_profile_enable", "ba_profile_reset", "ba_profile_get", "ba_dense_ldl_solve", "ba_dense_ldl_solve_f32",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.ba_last_error.restype = C.c_char_p
    vp, i64, f64 = C.c_void_p, C.c_int64, C.c_double
    L.ba_device_count.argtypes = [C.POINTER(C.c_int)]
    L.ba_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.ba_read_bal_header.argtypes = [C.c_char_p, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.ba_read_bal.argtypes = [C.c_char_p, i64, i64, i64, vp, vp, vp, vp]
    L.ba_read_bal_f32.argtypes = [C.c_char_p, i64, i64, i64, vp, vp, vp, vp]
    L.ba_problem_create.argtypes = [C.c_int, i64, i64, i64, vp, vp, vp, C.POINTER(vp)]
    L.ba_problem_destroy.argtypes = [vp]
    L.ba_problem_destroy.restype = None
    L.ba_problem_dims.argtypes = [vp] + [C.POINTER(i64)] * 6
    for name in ("ba_residual", "ba_residual_f32", "ba_jac_structure", "ba_jac_coord", "ba_jac_coord_f32"):
        getattr(L, name).argtypes = [vp, vp, vp]
    L.ba_jtr.argtypes = [vp, vp, vp, vp]
    for name in ("ba_residual_dev", "ba_residual_f32_dev", "ba_jac_structure_dev", "ba_jac_coord_dev",
                 "ba_jac_coord_f32_dev"):
        getattr(L, name).argtypes = [vp, vp, vp, vp]
    L.ba_jtr_dev.argtypes = [vp, vp, vp, vp, vp]
    L.ba_dev_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.ba_dev_free.argtypes = [vp, vp]
    L.ba_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.ba_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.ba_memcpy_h2d_on.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.ba_memcpy_d2h_on.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.ba_synchronize.argtypes = [vp]
    L.ba_lm_solve.argtypes = [vp, C.POINTER(LMOpts), vp, C.POINTER(LMStats), LOG_CB, vp]
    L.ba_lm_solve_dev.argtypes = [vp, C.POINTER(LMOpts), vp, C.POINTER(LMStats), LOG_CB, vp]
    L.ba_comm_get_unique_id.argtypes = [vp]
    L.ba_lm_set_comm_rccl.argtypes = [vp, C.c_int, C.c_int, vp]
    L.ba_lm_set_comm_hook.argtypes = [vp, C.c_int, C.c_int, COMM_CB, vp]
    L.ba_comm_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.ba_comm_stats_ops.argtypes = [vp, vp, vp]
    L.ba_dist_layout.argtypes = [i64, C.c_int, vp, vp]
    L.ba_lm_step.argtypes = [vp, vp, f64, vp, C.POINTER(f64), vp]
    L.ba_lm_step_f32.argtypes = [vp, vp, f64, vp, C.POINTER(f64), vp]
    L.ba_lm_step_pcg.argtypes = [vp, vp, f64, f64, C.c_int, vp, C.POINTER(f64), vp, C.POINTER(C.c_int)]
    L.ba_lm_schur_pattern.argtypes = [vp, C.POINTER(f64), C.POINTER(f64), C.POINTER(C.c_int)]
    L.ba_lm_schur_memory.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.ba_schur_ordering.argtypes = [i64, i64, i64, vp, vp, C.c_int, vp, C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]
    L.ba_lm_set_ordering.argtypes = [vp, C.c_int]
    L.ba_lm_schur_ordering.argtypes = [vp, vp, C.POINTER(C.c_char_p)]
    L.ba_profile_enable.argtypes = [vp, C.c_int]
    L.ba_profile_reset.argtypes = [vp]
    L.ba_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(f64), C.POINTER(i64), C.POINTER(C.c_int)]
    L.ba_dense_ldl_solve.argtypes = [C.c_int, i64, vp, vp, vp, C.POINTER(f64)]
    L.ba_dense_ldl_solve_f32.argtypes = [C.c_int, i64, vp, vp, vp, C.POINTER(f64)]
    _lib = L
    return L


def check(rc):
    if rc != BA_OK:
        msg = lib().ba_last_error().decode("utf-8", "replace")
        if rc == 4:
            raise SQDException(rc, msg)
        raise BAError(rc, msg)


def ptr(a):
    """host pointer of a C-contiguous numpy array"""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    try:
        rc = lib().ba_device_count(C.byref(n))
    except OSError:
        return 0
    return n.value if rc == BA_OK else 0


ORDERINGS = {"AMD": 0, "Metis": 1, "natural": 2}


def schur_ordering(cam_idx1, pnt_idx1, ncams, npnts, method="AMD"):
    """Fill-reducing camera ordering of a problem (`perm` of src/lm.jl:84-88 applied to the reduced camera system) and the
    tile fill it leaves: (perm1, tile_fill, flop_fill, block_fill); perm1[k] = 1-based camera at block row k of S.  Host
    only: needs no device."""
    cam = np.ascontiguousarray(cam_idx1, dtype=np.int64)
    pnt = np.ascontiguousarray(pnt_idx1, dtype=np.int64)
    perm = np.zeros(int(ncams), dtype=np.int64)
    tf, ff, bf = C.c_double(0), C.c_double(0), C.c_double(0)
    check(lib().ba_schur_ordering(int(ncams), int(npnts), len(cam), ptr(cam), ptr(pnt), ORDERINGS[method], ptr(perm),
                                  C.byref(tf), C.byref(ff), C.byref(bf)))
    return perm, tf.value, ff.value, bf.value


def dense_ldl_solve(A, b, device=0, f32=False):
    """Solve A x = b with the device blocked LDL' (A symmetric, only its lower triangle is read); f32: in Float32."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = A.shape[0]
    x = np.zeros(n)
    ms = C.c_double(0)
    fn = lib().ba_dense_ldl_solve_f32 if f32 else lib().ba_dense_ldl_solve
    check(fn(device, n, ptr(A), ptr(b), ptr(x), C.byref(ms)))
    return x, ms.value
