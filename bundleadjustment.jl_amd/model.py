"""Host-side mirror of the reference's model layer on top of the C ABI (include/ba_hip.h).

Reference interface mirrored (same names, argument meaning and error behaviour; Julia's `!` becomes a
trailing underscore because Python identifiers cannot contain it):

    BALNLPModel(filename, T)                      src/BALNLPModels.jl:79-106
    cons!(nlp, x, cx)                 -> cons_    src/BALNLPModels.jl:115-122
    jac_structure!(nlp, rows, cols)   -> jac_structure_   :125-158
    jac_coord!(nlp, x, vals)          -> jac_coord_       :161-206
    obj / grad!                                   :109-112
    FeasibilityResidual(nlp)  (NLPModels 0.12.4 adapter used at src/solve_ba.jl:25):
        residual!, jac_structure_residual!, jac_coord_residual!, nls_meta.nequ / nnzj

All arrays at this level are the reference's: 1-based int64 indices, x = [points; cameras], camera
(r, t, k1, k2, f).  Julia is not installed in this image, so this Python layer stands where the Julia shim of
INTEGRATION.md would; both are thin argument marshalling over the same C entry points.
"""
import ctypes as C
import sys
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .readfiles import name as _ba_name
from .readfiles import readfile


@dataclass
class NLPModelMeta:
    nvar: int
    ncon: int
    x0: np.ndarray
    lcon: np.ndarray
    ucon: np.ndarray
    nnzj: int
    name: str


@dataclass
class NLSMeta:
    nequ: int
    nvar: int
    nnzj: int
    x0: np.ndarray


@dataclass
class Counters:
    neval_cons: int = 0
    neval_jac: int = 0
    neval_residual: int = 0
    neval_jac_residual: int = 0


class BALNLPModel:
    """minimize 0 subject to F(x) = 0, F the reprojection residuals (src/BALNLPModels.jl:71-106)."""

    def __init__(self, filename=None, T=np.float64, *, arrays=None, device=0, model_name=None):
        T = np.dtype(T).type
        if T not in (np.float64, np.float32):
            raise TypeError("BALNLPModel supports Float64 and Float32")
        if arrays is None:
            cams_indices, pnts_indices, pt2d, x0, ncams, npnts, nobs = readfile(filename, T)
            model_name = _ba_name(filename) if model_name is None else model_name
        else:
            cams_indices, pnts_indices, pt2d, x0, ncams, npnts, nobs = arrays
            model_name = model_name or "BAL-arrays"
        self.T = T
        self.cams_indices = np.ascontiguousarray(cams_indices, dtype=np.int64)
        self.pnts_indices = np.ascontiguousarray(pnts_indices, dtype=np.int64)
        self.pt2d = np.ascontiguousarray(pt2d, dtype=T)
        self.nobs, self.npnts, self.ncams = int(nobs), int(npnts), int(ncams)
        nvar = 9 * self.ncams + 3 * self.npnts  # :95
        ncon = 2 * self.nobs                    # :97
        x0 = np.ascontiguousarray(x0, dtype=T)
        if x0.shape != (nvar,) or self.pt2d.shape != (ncon,) or self.cams_indices.shape != (self.nobs,) \
                or self.pnts_indices.shape != (self.nobs,):
            raise ValueError("BALNLPModel: array sizes do not match (ncams, npnts, nobs)")
        # lcon/ucon are Float64 zeros whatever T is (fill(0.0, ncon), :102)
        self.meta = NLPModelMeta(nvar=nvar, ncon=ncon, x0=x0, lcon=np.zeros(ncon), ucon=np.zeros(ncon),
                                 nnzj=2 * self.nobs * 12, name=model_name)
        self.counters = Counters()
        self._h = C.c_void_p()
        pt2d64 = np.ascontiguousarray(self.pt2d, dtype=np.float64)
        _lib.check(_lib.lib().ba_problem_create(device, self.ncams, self.npnts, self.nobs,
                                                _lib.ptr(self.cams_indices), _lib.ptr(self.pnts_indices),
                                                _lib.ptr(pt2d64), C.byref(self._h)))
        self.device = device

    # -- lifetime -------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().ba_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():  # at interpreter exit the HIP context may already be gone: leak instead
            return
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- NLPModels API ----------------------------------------------------------------------------------------
    def obj(self, x):
        return 0.0  # :109

    def grad_(self, x, g):
        g[:] = 0  # :112
        return g

    def _check_x(self, x):
        x = np.ascontiguousarray(x, dtype=self.T)
        if x.shape != (self.meta.nvar,):
            raise ValueError(f"x has length {x.shape}, expected {self.meta.nvar}")
        return x

    def cons_(self, x, cx):
        """cons!(nlp, x, cx): residuals at x, minus pt2d (NaN is kept, :119-120 is commented out in the reference)."""
        self.counters.neval_cons += 1
        x = self._check_x(x)
        if not (isinstance(cx, np.ndarray) and cx.dtype == self.T and cx.flags["C_CONTIGUOUS"]
                and cx.shape == (self.meta.ncon,)):
            raise ValueError("cx must be a contiguous vector of length ncon and the model's element type")
        fn = _lib.lib().ba_residual if self.T is np.float64 else _lib.lib().ba_residual_f32
        _lib.check(fn(self._h, _lib.ptr(x), _lib.ptr(cx)))
        return cx

    def cons(self, x):
        return self.cons_(x, np.empty(self.meta.ncon, dtype=self.T))

    def jac_structure_(self, rows, cols):
        """jac_structure!(nlp, rows, cols): 1-based COO pattern, 24 entries per observation."""
        self.counters.neval_jac += 1  # the reference bumps :neval_jac here too (:126)
        for a in (rows, cols):
            if not (isinstance(a, np.ndarray) and a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
                    and a.shape == (self.meta.nnzj,)):
                raise ValueError("rows/cols must be contiguous int64 vectors of length nnzj")
        _lib.check(_lib.lib().ba_jac_structure(self._h, _lib.ptr(rows), _lib.ptr(cols)))
        return rows, cols

    def jac_structure(self):
        return self.jac_structure_(np.empty(self.meta.nnzj, dtype=np.int64), np.empty(self.meta.nnzj, dtype=np.int64))

    def jac_coord_(self, x, vals):
        """jac_coord!(nlp, x, vals): 2x12 block per observation, row-major, NaN -> 0."""
        self.counters.neval_jac += 1
        x = self._check_x(x)
        if not (isinstance(vals, np.ndarray) and vals.dtype == self.T and vals.flags["C_CONTIGUOUS"]
                and vals.shape == (self.meta.nnzj,)):
            raise ValueError("vals must be a contiguous vector of length nnzj and the model's element type")
        fn = _lib.lib().ba_jac_coord if self.T is np.float64 else _lib.lib().ba_jac_coord_f32
        _lib.check(fn(self._h, _lib.ptr(x), _lib.ptr(vals)))
        return vals

    def jac_coord(self, x):
        return self.jac_coord_(x, np.empty(self.meta.nnzj, dtype=self.T))

    def jtprod_coo(self, vals, r):
        """J' r from COO values: mul_sparse(cols, rows, vals, r, nnzj, nvar) as called at src/lm.jl:57."""
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        out = np.empty(self.meta.nvar)
        _lib.check(_lib.lib().ba_jtr(self._h, _lib.ptr(vals), _lib.ptr(r), _lib.ptr(out)))
        return out

    # -- per-kernel timing ---------------------------------------------------------------------------------------
    def profile(self, on=True):
        _lib.check(_lib.lib().ba_profile_enable(self._h, int(on)))
        _lib.check(_lib.lib().ba_profile_reset(self._h))

    def profile_get(self):
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        calls = (C.c_int64 * cap)()
        n = C.c_int(0)
        _lib.check(_lib.lib().ba_profile_get(self._h, cap, names, ms, calls, C.byref(n)))
        return {names[i].decode(): (ms[i], calls[i]) for i in range(n.value)}


class FeasibilityResidual:
    """NLPModels.FeasibilityResidual(nlp): the NLS view F(x) = c(x) - lcon used by the LM drivers
    (src/solve_ba.jl:25, src/main.jl:27).  lcon = 0 here, so residual! is cons!."""

    def __init__(self, nlp):
        self.nlp = nlp
        m = nlp.meta
        self.meta = NLPModelMeta(nvar=m.nvar, ncon=0, x0=m.x0, lcon=np.zeros(0), ucon=np.zeros(0), nnzj=0,
                                 name=m.name + "-feasres")
        self.nls_meta = NLSMeta(nequ=m.ncon, nvar=m.nvar, nnzj=m.nnzj, x0=m.x0)
        self.counters = nlp.counters

    def residual_(self, x, Fx):
        self.counters.neval_residual += 1
        self.nlp.cons_(x, Fx)
        return Fx

    def residual(self, x):
        return self.residual_(x, np.empty(self.nls_meta.nequ, dtype=self.nlp.T))

    def jac_structure_residual_(self, rows, cols):
        return self.nlp.jac_structure_(rows, cols)

    def jac_structure_residual(self):
        return self.nlp.jac_structure()

    def jac_coord_residual_(self, x, vals):
        self.counters.neval_jac_residual += 1
        return self.nlp.jac_coord_(x, vals)

    def jac_coord_residual(self, x):
        return self.jac_coord_residual_(x, np.empty(self.nls_meta.nnzj, dtype=self.nlp.T))
