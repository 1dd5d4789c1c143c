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


REF_LOG = "/root/reference/benchmark/third/lm_linesearch.log"


@pytest.mark.skipif(not os.path.exists(REF_LOG), reason="the reference is mounted in the build container only")
def test_logtrace_parses_reference_benchmark_logs(ba):
    """benchmark/third/lm_linesearch.log:2-105 -- the published LadyBug-49 run of lm.jl: 57 iterations, final objective
    13364.31090775441.  The parser must recover the table, the parameters and the execution stats; the comparator must
    accept the run against itself and flag a flipped accept/reject or a changed digit."""
    from importlib import import_module
    lt = import_module(ba.__name__ + ".logtrace")
    runs = lt.parse_log(open(REF_LOG, encoding="utf-8").read())
    assert len(runs) == 22
    r = runs[0]
    assert r.problem == "LadyBug-49-7776-feasres" and r.first_line == 1
    assert r.params["facto"] == "LDL" and r.params["perm"] == "AMD" and r.params["linesearch"] is False and r.params["λ"] == 30.0
    assert r.tolerances["rtol"] == 6.055454452393343e-6
    assert len(r.rows) == 57 and r.iterations == 57 and r.rows[-1][0] == 57
    assert r.objective == 13364.31090775441 and r.status == "acceptable" and r.dual_feas == 160.33170664306257
    assert r.rows[0] == (1, 8.5e5, 0.0, 2.4e7, 4.2e2, 2.1, 1.0, True)
    assert r.rows[5][7] is False and r.rows[5][6] == -0.19
    assert abs(r.elapsed_time - 43.60739994049072) < 1e-12
    # comparator: against itself
    ok = lt.compare_trace(r.rows, r, objective=r.objective)
    assert ok["ok"] and ok["rows_compared"] == 57
    bad = [list(x) for x in r.rows]
    bad[9][7] = not bad[9][7]
    res = lt.compare_trace([tuple(x) for x in bad], r)
    assert not res["ok"] and res["first_mismatch"]["row"] == 9
    bad = [list(x) for x in r.rows]
    bad[3][4] *= 1.2
    assert lt.compare_trace([tuple(x) for x in bad], r)["first_mismatch"]["row"] == 3
    assert not lt.compare_trace(r.rows, r, objective=r.objective * (1 + 1e-5))["ok"]
    # a full-precision run prints as its 2-digit row
    full = [(it, f * 1.004, df * 0.996, g * 1.003, lam * 0.997, d * 1.002, rho, acc) for it, f, df, g, lam, d, rho, acc in r.rows]
    assert lt.compare_trace(full, r)["rows_compared"] == 57


def test_logtrace_roundtrip_of_a_device_style_log(ba):
    """rows as lm.py returns them -> the reference's text format -> parse -> compare."""
    from importlib import import_module
    lt = import_module(ba.__name__ + ".logtrace")
    rows = [(1, 851234.5, 0.0, 2.41e7, 417.3, 2.13, 0.998, True), (2, 21456.7, 829777.8, 5.4e5, 46.4, 5.06, -0.19, False)]
    text = "┌ Info: FeasibilityResidual - x\n│   Problem name: T-feasres\n[ Info:   iter      f(x)        Δf     ‖Jᵀr‖         λ       ‖δ‖         ρ           status  \n"
    text = text
    for it, f, df, g, lam, d, rho, acc in rows:
        text += "[ Info: %6d  %8.1e  %8.1e  %8.1e  %8.1e  %8.1e  %8.1e  %15s\n" % (it, f, df, g, lam, d, rho, "acc" if acc else "rej")
    text += "┌ Info: Generic Execution stats\n│   status: first-order stationary\n│   objective value: 21456.7\n│   iterations: 2\n└   elapsed time: 1.5\n"
    runs = lt.parse_log(text)
    assert len(runs) == 1 and len(runs[0].rows) == 2 and runs[0].status == "first_order" and runs[0].objective == 21456.7
    assert lt.compare_trace(rows, runs[0], objective=21456.7)["ok"]


@pytest.mark.parametrize("ext", [".txt", ".txt.bz2"])
def test_reader_streams_through_a_small_window(ba, small_prob, tmp_path, ext, monkeypatch):
    """The reader decodes the file chunk by chunk into a fixed window (8 MiB) and parses as it goes; with a 4 KiB window
    a 300 KB file is refilled ~80 times and every token still parses identically (no number straddles a refill)."""
    p = small_prob
    path = str(tmp_path / "Synth" / ("problem-12-400-pre" + ext))
    ba.synthetic.write_bal(path, p)
    monkeypatch.setenv("BA_READ_CACHE", "0")
    monkeypatch.setenv("BA_READER_WINDOW", "4096")
    cam, pnt, pt2d, x0, ncams, npnts, nobs = ba.readfile(path)
    assert (ncams, npnts, nobs) == (p["ncams"], p["npnts"], p["nobs"])
    assert np.array_equal(cam, p["cam_idx1"]) and np.array_equal(pnt, p["pnt_idx1"])
    assert np.array_equal(pt2d, p["pt2d"]) and np.array_equal(x0, p["x0"])
    # truncated in the middle of the camera block: refused, not silently short
    data = open(path, "rb").read() if ext == ".txt" else None
    if data is not None:
        cut = str(tmp_path / "cut.txt")
        with open(cut, "wb") as fh:
            fh.write(data[: len(data) // 2])
        with pytest.raises(ba.BAError):
            ba.readfile(cut)


def test_reader_cache_corrupt_or_foreign_is_ignored(ba, small_prob, tmp_path):
    """A truncated cache (BadZipFile), a cache with the wrong dtypes, and one made from another source file must all be
    ignored and the file parsed again."""
    p = small_prob
    path = str(tmp_path / "Synth" / "problem-12-400-pre.txt")
    ba.synthetic.write_bal(path, p)
    a = ba.readfile(path)
    cache = path + ".f64.balcache.npz"
    blob = open(cache, "rb").read()
    os.utime(path, (1, 1))  # keep the source older than whatever cache is written below
    # source stamp changed -> the cache belongs to another file
    assert ba.readfiles._cache_load(cache, path, np.float64) is None
    ba.readfile(path)  # rewrites the cache for the new stamp
    assert ba.readfiles._cache_load(cache, path, np.float64) is not None
    with open(cache, "wb") as fh:
        fh.write(blob[: len(blob) // 3])
    b = ba.readfile(path)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    z = dict(np.load(cache))
    z["x0"] = z["x0"].astype(np.float32)
    np.savez(cache[:-4], **z)
    assert ba.readfiles._cache_load(cache, path, np.float64) is None
    z["x0"] = z["x0"].astype(np.float64)
    z["pnt"] = z["pnt"].astype(np.int32)
    np.savez(cache[:-4], **z)
    assert ba.readfiles._cache_load(cache, path, np.float64) is None


def test_c_abi_from_plain_c(ba, small_prob, tmp_path):
    """include/ba_hip.h compiles as C99 (-pedantic -Werror), libba_hip.so links from C, and the struct layouts the
    foreign-function bindings mirror by hand (ctypes here, the Julia structs in julia/) equal the C compiler's."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    libdir = os.path.dirname(ba._lib.LIB_PATH)
    exe = str(tmp_path / "abi_check")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-L", libdir, "-lba_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bal = str(tmp_path / "Synth" / "problem-12-400-pre.txt.bz2")
    ba.synthetic.write_bal(bal, small_prob)
    r = subprocess.run([exe, bal], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    sizes, offs = {}, {}
    for line in r.stdout.splitlines():
        w = line.split()
        if w[0] == "sizeof":
            sizes[w[1]] = int(w[2])
        elif w[0] == "offsetof":
            offs[(w[1], w[2])] = int(w[3])
    for cname, cls in (("ba_lm_opts", ba._lib.LMOpts), ("ba_lm_stats", ba._lib.LMStats)):
        assert sizes[cname] == ctypes.sizeof(cls)
        names = [f[0] for f in cls._fields_]
        c_names = [k[1] for k in offs if k[0] == cname]
        assert len(names) == len(c_names)
        for fname in names:
            c_field = {"lam": "lambda"}.get(fname, fname)
            assert offs[(cname, c_field)] == getattr(cls, fname).offset, (cname, fname)
    p = small_prob
    assert f"ncams {p['ncams']} npnts {p['npnts']} nobs {p['nobs']}" in r.stdout
    assert "ba_read_bal rc 0" in r.stdout and "ba_problem_dims(NULL) rc 1" in r.stdout


def test_product_library_has_no_probe_entries(ba):
    """micro-benchmark / probe entry points live in tools/libba_bench.so (csrc/bench/), not in the product library"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", ba._lib.LIB_PATH], capture_output=True, text=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert not [s for s in syms if s.startswith("ba_debug")]
    exported = {s for s in syms if s.startswith("ba_")} - {"ba_set_error"}
    assert exported == set(ba._lib.SYMBOLS), exported ^ set(ba._lib.SYMBOLS)


def test_julia_shim_matches_header(ba):
    """julia/BALHIP.jl cannot be executed here (no Julia in the image): its struct field lists are compared with the
    ctypes mirrors, which test_c_abi_from_plain_c ties to the C compiler's layout; every ccall symbol must be exported."""
    src = {f: open(os.path.join(ROOT, "julia", f), encoding="utf-8").read()
           for f in ("BALHIP.jl", "BALNLPModelsHIP.jl", "LevenbergMarquardtHIP.jl", "solve_ba_hip.jl")}
    jl_types = {"Cint": ctypes.c_int, "Cdouble": ctypes.c_double}

    def fields(struct):
        body = re.search(r"struct " + struct + r"\n(.*?)\n(?:  " + struct + r"\(\)|end)", src["BALHIP.jl"], re.S).group(1)
        return [(m.group(1), jl_types[m.group(2)]) for m in re.finditer(r"^\s+(\w+) :: (\w+)\s*$", body, re.M)]

    for jname, cls in (("BaLmOpts", ba._lib.LMOpts), ("BaLmStats", ba._lib.LMStats)):
        got = fields(jname)
        want = [({"lam": "lambda"}.get(n, n), t) for n, t in cls._fields_]
        assert got == want, (jname, got, want)
    called = set(re.findall(r"ccall\(\(:(\w+), libba\)", "".join(src.values())))
    assert called and called <= set(ba._lib.SYMBOLS), called - set(ba._lib.SYMBOLS)
    # both reference signatures exist: 5 positional arguments (src/lm.jl:15-19) and 4 (src/LevenbergMarquardt.jl:16-19)
    lm = src["LevenbergMarquardtHIP.jl"]
    assert re.search(r"function Levenberg_Marquardt\(model :: AbstractNLSModel, facto :: Symbol, perm :: Symbol, normalize :: Symbol,\s+linesearch :: Bool;", lm)
    assert re.search(r"function Levenberg_Marquardt\(model :: AbstractNLSModel, facto :: Symbol, perm :: Symbol, normalize :: Symbol;", lm)
    for needed in ("NLPModels.cons!", "NLPModels.jac_structure!", "NLPModels.jac_coord!", "function name(", "function readfile(",
                   "Vector{Float32}"):
        assert needed in src["BALNLPModelsHIP.jl"]
