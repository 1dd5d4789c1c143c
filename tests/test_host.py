"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/ba_hip.h declares,
the BAL reader (plain and bzip2) round-trips, name(), sharding helpers, and the product path refuses to run without
a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(ba):
    hdr = open(os.path.join(ROOT, "include", "ba_hip.h")).read()
    declared = set(re.findall(r"\b(ba_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ba_allreduce_fn", "ba_log_cb"}
    L = ba._lib.lib()
    for s in sorted(declared):
        assert hasattr(L, s), f"libba_hip.so lacks {s}"
    assert declared == set(ba._lib.SYMBOLS)


def test_no_cpu_fallback_without_device(ba):
    if ba.device_count() > 0:
        pytest.skip("a device is visible here")
    with pytest.raises(ba.BAError):
        ba.BALNLPModel(arrays=(np.array([1], np.int64), np.array([1], np.int64), np.zeros(2), np.zeros(12), 1, 1, 1))


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "bundleadjustment.jl_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "ba_oracle" not in txt and "libba_oracle" not in txt, fn


@pytest.mark.parametrize("ext", [".txt", ".txt.bz2"])
@pytest.mark.parametrize("T", [np.float64, np.float32])
def test_reader_roundtrip(ba, small_prob, tmp_path, ext, T):
    p = small_prob
    path = str(tmp_path / "Synth" / ("problem-12-400-pre" + ext))
    ba.synthetic.write_bal(path, p)
    cam, pnt, pt2d, x0, ncams, npnts, nobs = ba.readfile(path, T)
    assert (ncams, npnts, nobs) == (p["ncams"], p["npnts"], p["nobs"])
    assert cam.dtype == np.int64 and np.array_equal(cam, p["cam_idx1"]) and np.array_equal(pnt, p["pnt_idx1"])
    assert pt2d.dtype == T and x0.dtype == T
    # parse(T, str) rounds the decimal once: Float32 values equal the Float64 ones rounded
    assert np.array_equal(pt2d, p["pt2d"].astype(T)) and np.array_equal(x0, p["x0"].astype(T))


def test_reader_binary_cache(ba, small_prob, tmp_path, monkeypatch):
    """The parsed arrays are cached beside the file and reused until the file changes; BA_READ_CACHE=0 disables."""
    import time
    p = small_prob
    path = str(tmp_path / "Synth" / "problem-12-400-pre.txt.bz2")
    ba.synthetic.write_bal(path, p)
    a = ba.readfile(path)
    cache = path + ".f64.balcache.npz"
    assert os.path.exists(cache)
    b = ba.readfile(path)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    # a stale cache (source rewritten with different content) is ignored
    q = dict(p)
    q["pt2d"] = p["pt2d"] + 1.0
    time.sleep(0.05)
    ba.synthetic.write_bal(path, q)
    os.utime(path, (time.time() + 5, time.time() + 5))
    c = ba.readfile(path)
    assert np.allclose(c[2], q["pt2d"]) and not np.allclose(c[2], a[2])
    monkeypatch.setenv("BA_READ_CACHE", "0")
    os.remove(cache) if os.path.exists(cache) else None
    ba.readfile(path)
    assert not os.path.exists(cache)


def test_reader_camera_order(ba, tmp_path):
    # file order r t f k1 k2 -> stored r t k1 k2 f (src/ReadFiles.jl:32-43)
    path = str(tmp_path / "one.txt")
    vals = [0.1, 0.2, 0.3, 1, 2, 3, 500.0, -1e-7, 2e-12]
    with open(path, "w") as fh:
        fh.write("1 1 1\n0 0 10.5 -3.25\n" + "\n".join(repr(v) for v in vals) + "\n7\n8\n9\n")
    cam, pnt, pt2d, x0, ncams, npnts, nobs = ba.readfile(path)
    assert cam[0] == 1 and pnt[0] == 1 and list(pt2d) == [10.5, -3.25]
    assert list(x0) == [7, 8, 9, 0.1, 0.2, 0.3, 1, 2, 3, -1e-7, 2e-12, 500.0]


def test_reader_errors(ba, tmp_path):
    with pytest.raises(ba.BAError):
        ba.readfile(str(tmp_path / "missing.txt"))
    path = str(tmp_path / "trunc.txt")
    with open(path, "w") as fh:
        fh.write("1 1 1\n0 0 1.0 2.0\n0.1\n")
    with pytest.raises(ba.BAError):
        ba.readfile(path)


def test_name(ba):
    assert ba.name("LadyBug/problem-49-7776-pre.txt.bz2") == "LadyBug-49-7776"  # src/BALNLPModels.jl:58-68
    assert ba.name("Dubrovnik/problem-356-226730-pre.txt.bz2") == "Dubrovnik-356-226730"


def test_synthetic_generator(ba):
    p = ba.synthetic.make_named("ladybug-49")
    assert (p["ncams"], p["npnts"], p["nobs"]) == (49, 7776, 31843)
    assert np.all(np.diff(p["pnt_idx1"]) >= 0)  # BAL order: grouped by point
    key = p["pnt_idx1"] * 100000 + p["cam_idx1"]
    assert np.all(np.diff(key) > 0)  # ... then by camera, no duplicates
    deg = np.bincount(p["pnt_idx1"] - 1, minlength=p["npnts"])
    assert deg.min() >= 2 and deg.sum() == p["nobs"]
    q = ba.synthetic.make_named("ladybug-49")
    assert np.array_equal(p["x0"], q["x0"]) and np.array_equal(p["pt2d"], q["pt2d"])  # seeded


def test_partition_by_point(ba, small_prob):
    p = small_prob
    for world in (1, 2, 3, 8):
        parts = ba.parallel.partition_by_point(p["pnt_idx1"], p["npnts"], world)
        assert parts[0][0] == 0 and parts[-1][1] == p["npnts"]
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        tot = 0
        for r in range(world):
            arrs, info = ba.parallel.shard_problem(ba.synthetic.as_arrays(p), r, world)
            tot += arrs[6]
            assert arrs[1].min() >= 1 and arrs[1].max() <= arrs[5]
            assert len(arrs[3]) == 3 * arrs[5] + 9 * p["ncams"]
            assert abs(arrs[6] - p["nobs"] / world) <= 0.1 * p["nobs"] + 64
        assert tot == p["nobs"]


def test_jac_kernel_inflight_registers_untouched():
    """k_jac_coord requests its camera rows with inline-asm loads the compiler cannot track; the generated code must
    not touch their destination registers before the matching inline-asm s_waitcnt (tools/check_jac_isa.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_jac_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
