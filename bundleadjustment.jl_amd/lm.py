"""Levenberg_Marquardt(model, facto, perm, normalize[, linesearch]; kwargs...) -- host mirror of the two solver
entry points of the reference over ba_lm_solve (include/ba_hip.h):

  * 4 positional arguments  -> src/LevenbergMarquardt.jl:16-26 (old API; what src/solve_ba.jl:26 calls)
  * 5 positional arguments  -> src/lm.jl:15-26 (new API with `linesearch`; src/main.jl:30, src/diffprecsions.jl:41)

Keyword names follow the reference with ASCII spellings (νd -> nu_d, νm -> nu_m, λ -> lam, δd -> delta_d).
Returns a GenericExecutionStats with the fields the reference fills (src/lm.jl:409-415,
src/LevenbergMarquardt.jl:384).
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .model import BALNLPModel, FeasibilityResidual

_FACTO = {"LDL": 0, "QR": 1, "PCG": 2}
_NORM = {"None": 0, "J": 1, "A": 2}
_PERM = ("AMD", "Metis", "natural")


@dataclass
class GenericExecutionStats:
    status: str
    solution: np.ndarray
    objective: float
    iter: int
    elapsed_time: float
    dual_feas: float = float("inf")
    primal_feas: float = float("inf")
    # extras (not in the reference struct)
    loop_time: float = 0.0
    n_accepted: int = 0
    n_rejected: int = 0
    n_residual: int = 0
    n_jacobian: int = 0
    n_factor: int = 0
    lambda_final: float = 0.0
    log: list = field(default_factory=list)
    n_cg: int = 0  # facto = :PCG: conjugate-gradient iterations over the whole solve

    def __str__(self):
        return (f"Generic Execution stats\n  status: {self.status}\n  objective value: {self.objective!r}\n"
                f"  primal feasibility: {self.primal_feas!r}\n  dual feasibility: {self.dual_feas!r}\n"
                f"  solution: [{'  '.join(repr(float(v)) for v in self.solution[:4])} ⋯ {float(self.solution[-1])!r}]\n"
                f"  iterations: {self.iter}\n  elapsed time: {self.elapsed_time!r}")


def _sym(s):
    return s[1:] if isinstance(s, str) and s.startswith(":") else s


def Levenberg_Marquardt(model, facto, perm, normalize, linesearch=None, *, x=None, facto_type=None,
                        restol=None, satol=None, srtol=None, oatol=None, ortol=None, atol=None, rtol=None,
                        nu_d=None, nu_m=None, lam=None, delta_d=None, ite_max=None, max_time=None, verbose=False,
                        log=True, pcg_tol=None, pcg_max_iter=None, x_device_ptr=None):
    """x_device_ptr (an extension for device-resident callers, e.g. bench.py): the address of nvar doubles of DEVICE memory
    holding x0; the loop then runs through ba_lm_solve_dev -- no host copy of the iterate on either side -- the solution
    stays there and `solution` of the result is None."""
    facto, perm, normalize = _sym(facto), _sym(perm), _sym(normalize)
    if facto not in _FACTO:
        raise ValueError(f"facto must be :QR, :LDL or :PCG (extension: matrix-free CG on the reduced camera system), got {facto!r}")
    if perm not in _PERM:
        raise ValueError(f"perm must be :AMD or :Metis (or :natural, an extension: the caller's camera numbering), got {perm!r}")
    if normalize not in _NORM:
        raise ValueError(f"normalize must be :None, :J or :A, got {normalize!r}")
    nlp = model.nlp if isinstance(model, FeasibilityResidual) else model
    if not isinstance(nlp, BALNLPModel):
        raise TypeError("model must be a FeasibilityResidual(BALNLPModel) or a BALNLPModel")
    xf32 = nlp.T is np.float32  # eltype(x) = Float32: facto_type defaults to it (lm.jl:20), eps(T) tolerances
    variant = 0 if linesearch is None else 1
    if variant == 0 and (facto_type is not None or max_time is not None):
        raise TypeError("LevenbergMarquardt.jl's Levenberg_Marquardt has no facto_type / max_time keyword")
    if x_device_ptr is not None and x is not None:
        raise TypeError("x and x_device_ptr are exclusive")
    x0 = None if x_device_ptr is not None else np.array(nlp.meta.x0 if x is None else x, dtype=np.float64, copy=True)
    if x0 is not None and x0.shape != (nlp.meta.nvar,):
        raise ValueError("x has the wrong length")
    if facto_type is not None and np.dtype(facto_type) not in (np.dtype(np.float64), np.dtype(np.float32), np.dtype(np.float16)):
        raise TypeError("facto_type must be Float64, Float32 or Float16")
    # ba_lm_opts.facto_type: 0 = eltype(x), 1 = Float32, 2 = Float16 (src/lm.jl:165-173)
    if facto_type is None:
        ft = 1 if (xf32 and variant == 1) else 0  # facto_type defaults to eltype(x) (lm.jl:20)
    else:
        ft = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.float16): 2}[np.dtype(facto_type)]
    if ft == 2 and _FACTO[facto] != 0:
        raise ValueError("facto_type = Float16 exists in the :LDL branch only (src/lm.jl:92-95)")
    if facto == "PCG" and normalize != "None":
        raise ValueError("facto = :PCG has its own scaling (block-Jacobi preconditioner): normalize must be :None")
    if facto == "PCG" and facto_type is not None and ft == 1 and not xf32:
        raise ValueError("facto = :PCG runs in Float64: facto_type = Float32 belongs to the direct branches")

    def d(v):
        return -1.0 if v is None else float(v)

    o = _lib.LMOpts(variant=variant, facto=_FACTO[facto], normalize=_NORM[normalize], linesearch=int(bool(linesearch)),
                    facto_type=ft, ite_max=-1 if ite_max is None else int(ite_max), verbose=int(verbose), x_f32=int(xf32),
                    restol=d(restol), satol=d(satol), srtol=d(srtol), oatol=d(oatol), ortol=d(ortol), atol=d(atol),
                    rtol=d(rtol), nu_d=d(nu_d), nu_m=d(nu_m), lam=d(lam), delta_d=d(delta_d), max_time=d(max_time),
                    pcg_tol=d(pcg_tol), pcg_max_iter=-1 if pcg_max_iter is None else int(pcg_max_iter),
                    perm=_lib.ORDERINGS[perm])  # src/lm.jl:84-88: orders the cameras of the reduced system (ba_order.cpp)
    st = _lib.LMStats()
    rows = []

    def _cb(ctx, it, f, df, njtr, lmb, nd, rho, acc):
        rows.append((it, f, df, njtr, lmb, nd, rho, bool(acc)))

    cb = _lib.LOG_CB(_cb) if log else C.cast(None, _lib.LOG_CB)
    if x_device_ptr is not None:
        _lib.check(_lib.lib().ba_lm_solve_dev(nlp.handle, C.byref(o), C.c_void_p(int(x_device_ptr)), C.byref(st), cb, None))
    else:
        _lib.check(_lib.lib().ba_lm_solve(nlp.handle, C.byref(o), _lib.ptr(x0), C.byref(st), cb, None))
    nlp.counters.neval_cons += st.n_residual
    nlp.counters.neval_residual += st.n_residual
    nlp.counters.neval_jac += st.n_jacobian + 1  # + jac_structure! (src/BALNLPModels.jl:126)
    nlp.counters.neval_jac_residual += st.n_jacobian
    out = GenericExecutionStats(status=_lib.STATUS[st.status], solution=None if x0 is None else (x0.astype(nlp.T) if xf32 else x0), objective=st.objective, iter=st.iter,
                                elapsed_time=st.elapsed_s, loop_time=st.loop_s, n_accepted=st.n_accepted,
                                n_rejected=st.n_rejected, n_residual=st.n_residual, n_jacobian=st.n_jacobian,
                                n_factor=st.n_factor, lambda_final=st.lambda_final, log=rows, n_cg=st.n_cg)
    if variant == 1:
        out.dual_feas = st.dual_feas    # lm.jl:415
    else:
        out.primal_feas = st.dual_feas  # LevenbergMarquardt.jl:384 passes |J'r| as primal_feas
    return out


def lm_step(nlp, x, lam, want_jtr=True, facto_type=None, pcg=None):
    """One linear LM step from (x, lambda): delta, 1/2|J delta + r|^2, J'r  (ba_lm_step; facto_type=np.float32:
    ba_lm_step_f32, the reduced camera system factored in Float32 as src/lm.jl:170-173 does; pcg=(tol, max_iter):
    ba_lm_step_pcg, the step by preconditioned CG -- the CG iteration count is then appended to the result)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    delta = np.empty(nlp.meta.nvar)
    jtr = np.empty(nlp.meta.nvar) if want_jtr else None
    half = C.c_double(0)
    if pcg is not None:
        its = C.c_int(0)
        _lib.check(_lib.lib().ba_lm_step_pcg(nlp.handle, _lib.ptr(x), float(lam), float(pcg[0]), int(pcg[1]), _lib.ptr(delta),
                                             C.byref(half), _lib.ptr(jtr) if want_jtr else None, C.byref(its)))
        return delta, half.value, jtr, its.value
    f32 = facto_type is not None and np.dtype(facto_type) == np.float32
    fn = _lib.lib().ba_lm_step_f32 if f32 else _lib.lib().ba_lm_step
    _lib.check(fn(nlp.handle, _lib.ptr(x), float(lam), _lib.ptr(delta), C.byref(half), _lib.ptr(jtr) if want_jtr else None))
    return delta, half.value, jtr


def schur_pattern(nlp):
    """(tile_fill, flop_fill, sparse_schedule) of the reduced camera system of a handle that has run a direct solve
    (ba_lm_schur_pattern): the fraction of the lower 128 x 128 tiles in the factor's pattern, the fraction of the dense
    factorisation's trailing-update tiles that pattern needs, and whether the block-sparse list schedule is in use."""
    tf, ff, sp = C.c_double(0), C.c_double(0), C.c_int(0)
    _lib.check(_lib.lib().ba_lm_schur_pattern(nlp.handle, C.byref(tf), C.byref(ff), C.byref(sp)))
    return tf.value, ff.value, bool(sp.value)


def set_ordering(nlp, perm):
    """Camera ordering of the handle's next direct solves outside Levenberg_Marquardt (lm_step): "AMD", "Metis", "natural"."""
    _lib.check(_lib.lib().ba_lm_set_ordering(nlp.handle, _lib.ORDERINGS[_sym(perm)]))


def schur_ordering_used(nlp):
    """(perm1, name): the camera sequence of the reduced camera system of a handle that has run a direct solve (perm1[k] =
    1-based camera at block row k of S) and the name of the candidate sequence that won (ba_lm_schur_ordering)."""
    perm = np.zeros(nlp.ncams, dtype=np.int64)
    name = C.c_char_p()
    _lib.check(_lib.lib().ba_lm_schur_ordering(nlp.handle, _lib.ptr(perm), C.byref(name)))
    return perm, name.value.decode()


def schur_memory(nlp):
    """(tiles_full, tiles_held, tiles_staging) of a handle that has run a direct solve (ba_lm_schur_memory): the 128 x 128
    tiles of the whole reduced camera matrix, what this handle holds of it, and its staging buffer (distributed runs)."""
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().ba_lm_schur_memory(nlp.handle, C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value
