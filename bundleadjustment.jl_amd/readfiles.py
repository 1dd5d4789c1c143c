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


def _cache_path(path, T):
    """Binary cache of a parsed BAL file: parsing is bound by single-stream bzip2 decompression
    (~18 MB/s of text: 5 s for Dubrovnik-356, ~100 s for Final-13682), so the parsed arrays are kept next to the data
    (or under BA_CACHE_DIR) and reused while they are newer than the source file.  BA_READ_CACHE=0 turns it off."""
    if os.environ.get("BA_READ_CACHE", "1") == "0":
        return None
    tag = "f64" if T is np.float64 else "f32"
    d = os.environ.get("BA_CACHE_DIR")
    if d:
        import hashlib
        return os.path.join(d, hashlib.sha1(os.path.abspath(path).encode()).hexdigest()[:16] + f".{tag}.balcache.npz")
    return path + f".{tag}.balcache.npz"


def _source_stamp(path):
    st = os.stat(path)
    return np.array([st.st_size, st.st_mtime_ns], dtype=np.int64)


def _cache_load(cpath, path, T):
    """the cached arrays, or None when there is no cache, it is older than / was made from another source file, it is
    truncated or corrupt, or its arrays are not what readfile(path, T) returns (shapes AND dtypes)"""
    import zipfile
    try:
        if cpath is None or os.path.getmtime(cpath) < os.path.getmtime(path):
            return None
        with np.load(cpath, allow_pickle=False) as z:
            dims = z["dims"]
            if "source" not in z.files or not np.array_equal(z["source"], _source_stamp(path)):
                return None
            out = (z["cam"], z["pnt"], z["pt2d"], z["x0"], int(dims[0]), int(dims[1]), int(dims[2]))
        ncams, npnts, nobs = out[4:]
        ok = (out[0].dtype == np.int64 and out[1].dtype == np.int64 and out[2].dtype == T and out[3].dtype == T
              and out[0].shape == (nobs,) and out[1].shape == (nobs,) and out[2].shape == (2 * nobs,)
              and out[3].shape == (3 * npnts + 9 * ncams,))
        return out if ok else None
    except (OSError, ValueError, KeyError, EOFError, zipfile.BadZipFile):
        return None


def readfile(filename, T=np.float64):
    """-> cam_indices, pnt_indices, pt2d, x0, ncams, npnts, nobs  (1-based indices, x0 = [points; cameras])."""
    T = np.dtype(T).type
    rpath = resolve(filename)
    cpath = _cache_path(rpath, T) if os.path.exists(rpath) else None
    hit = _cache_load(cpath, rpath, T)
    if hit is not None:
        return hit
    out = _parse(rpath, T)
    if cpath is not None:
        try:
            os.makedirs(os.path.dirname(cpath) or ".", exist_ok=True)
            tmp = cpath + f".tmp{os.getpid()}.npz"
            np.savez(tmp, cam=out[0], pnt=out[1], pt2d=out[2], x0=out[3], dims=np.array(out[4:7], dtype=np.int64),
                     source=_source_stamp(rpath))
            os.replace(tmp, cpath)
        except OSError:
            pass  # read-only data directory: parse every time
    return out


def _parse(rpath, T):
    path = rpath.encode()
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
