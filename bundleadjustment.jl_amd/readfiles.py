"""readfile(filename, T) and name(filename): host mirror of src/ReadFiles.jl:9-53 and
src/BALNLPModels.jl:58-68 over the C reader (ba_read_bal_header / ba_read_bal)."""
import ctypes as C
import os

import numpy as np

from . import _lib

# The reference resolves `filename` against <repo>/Data (ReadFiles.jl:10).  Same convention, overridable.
DATA_DIR = os.environ.get("BA_DATA_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "Data"))


def resolve(filename):
    if os.path.isabs(filename) or os.path.exists(filename):
        return filename
    return os.path.join(DATA_DIR, filename)


def readfile(filename, T=np.float64):
    """-> cam_indices, pnt_indices, pt2d, x0, ncams, npnts, nobs  (1-based indices, x0 = [points; cameras])."""
    T = np.dtype(T).type
    path = resolve(filename).encode()
    L = _lib.lib()
    nc, npt, no = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(L.ba_read_bal_header(path, C.byref(nc), C.byref(npt), C.byref(no)))
    ncams, npnts, nobs = nc.value, npt.value, no.value
    cam = np.empty(nobs, dtype=np.int64)
    pnt = np.empty(nobs, dtype=np.int64)
    pt2d = np.empty(2 * nobs, dtype=T)
    x0 = np.empty(3 * npnts + 9 * ncams, dtype=T)
    fn = L.ba_read_bal if T is np.float64 else L.ba_read_bal_f32
    _lib.check(fn(path, ncams, npnts, nobs, _lib.ptr(cam), _lib.ptr(pnt), _lib.ptr(pt2d), _lib.ptr(x0)))
    return cam, pnt, pt2d, x0, ncams, npnts, nobs


def name(filename):
    """"LadyBug/problem-49-7776-pre.txt.bz2" -> "LadyBug-49-7776"  (src/BALNLPModels.jl:58-68: text before the
    first '/', then from 8 characters after it up to 2 before the next 'p')."""
    k = filename.index("/")
    l = k + 8
    while filename[l] != "p":
        l += 1
    return filename[:k] + filename[k + 8: l - 1]
