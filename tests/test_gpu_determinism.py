"""Determinism of the device path (same input, same process, same schedule -> the same bits) and the multi-rank path on
ONE GPU in ONE process with genuinely asynchronous streams (tests/helpers/ba_loopback.hip: every rank is a handle driven by
its own host thread; reduce / broadcast are stream-ordered device copies, no host thread waits for the GPU).

Why these exist: round 2's driver run failed `test_dist_lookahead_equals_alternating_schedule` on one box and passed it on
another.  The host-staged hook copied with null-stream hipMemcpy while the look-ahead launches the consumers of those copies
on a second non-blocking stream the host had not synchronised ("What went wrong in round 2")."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from _util import bits_report, digest, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOPBACK = os.path.join(ROOT, "tests", "helpers", "libba_loopback.so")

pytestmark = pytest.mark.gpu


# ---- one process, one rank: the same call twice gives the same bits ------------------------------------------------------
@pytest.mark.parametrize("n,schedule", [(1800, "in order (15 tile rows)"), (4480, "hoisted diagonal tile (35 tile rows)"),
                                        (8100, "fused pair schedule (64 tile rows)")])
def test_dense_factorisation_twice_same_bits(ba, gpu_ok, n, schedule):
    """The dense blocked LDL' on the same matrix twice in one process: bit-equal solutions, for each of the three
    single-GPU schedules (the hoisted / fused ones run a workgroup BESIDE the trailing update and join it by a flag and an
    event: a missing fence there would show as run-to-run differences)."""
    rng = np.random.default_rng(n)
    R = rng.standard_normal((n, n))
    A = R + R.T
    del R
    A[np.diag_indices(n)] += 4 * np.sqrt(n)
    b = rng.standard_normal(n)
    x1, _ = ba._lib.dense_ldl_solve(A, b)
    x2, _ = ba._lib.dense_ldl_solve(A, b)
    res = np.linalg.norm(A @ x1 - b) / np.linalg.norm(b)
    assert res < 1e-12, f"n = {n} ({schedule}): residual {res:.3e}"
    rep = bits_report(x1, x2, f"dense LDL' solution, n = {n}, {schedule}, run 1 vs run 2")
    assert not rep, rep
    x3, _ = ba._lib.dense_ldl_solve(A, b, f32=True)
    x4, _ = ba._lib.dense_ldl_solve(A, b, f32=True)
    rep = bits_report(x3, x4, f"Float32 dense LDL' solution, n = {n}, {schedule}, run 1 vs run 2")
    assert not rep, rep


def test_lm_step_twice_same_bits_venice_scaled(ba, gpu_ok):
    """One ba_lm_step on a Venice-shaped problem at a quarter of its size (445 cameras: n = 4005, 32 tile rows; 1.25 M
    observations) twice on one handle, then on a second handle: residual, Jacobian, normal-equation blocks, Schur assembly
    (split keys + fixed-order combine), factorisation, solves and back-substitution give the same bits every time."""
    p = ba.synthetic.make_named("venice-1778", scale=0.25)
    arrays = ba.synthetic.as_arrays(p)
    m = ba.BALNLPModel(arrays=arrays)
    d1, h1, g1 = ba.lm_step(m, p["x0"], 30.0)
    d2, h2, g2 = ba.lm_step(m, p["x0"], 30.0)
    m.close()
    m = ba.BALNLPModel(arrays=arrays)
    d3, h3, g3 = ba.lm_step(m, p["x0"], 30.0)
    d32a, _, _ = ba.lm_step(m, p["x0"], 30.0, facto_type=np.float32)
    d32b, _, _ = ba.lm_step(m, p["x0"], 30.0, facto_type=np.float32)
    m.close()
    assert np.all(np.isfinite(d1))
    for tag, (a, b) in {"step, same handle": (d1, d2), "step, fresh handle": (d1, d3), "J'r, same handle": (g1, g2),
                        "J'r, fresh handle": (g1, g3), "Float32-factor step, same handle": (d32a, d32b)}.items():
        rep = bits_report(a, b, tag)
        assert not rep, rep
    assert h1 == h2 == h3, f"model value differs between runs: {h1!r} {h2!r} {h3!r}"



@pytest.mark.parametrize("ncams,shuffle", [(1100, False), (1100, True), (1095, False)])
def test_block_sparse_two_chains(ba, orc, gpu_ok, ncams, shuffle):
    """A profile eliminated from both ends (ba_order.cpp: "two-ended"): the list schedule runs the two independent runs of tile
    column pairs together, pair i of either run in the same launches (dense_ldl_factor_sparse, "two runs per launch"), and the
    remaining pairs in order.  1 100 cameras (n =
    9 900, 78 tile rows), cameras of a point within 10 % of the cameras, as generated and renumbered at random: the sequence
    chosen is the two-ended one, the pattern reports two runs, the step equals the oracle's (its ldl_analyse handed the same
    camera sequence) to 1e-9 and the one-chain schedule's (BA_SPARSE_TWO_RUNS=0) to 1e-11, Float32 to Float32 level; the
    same bits twice, and recorded graphs = plain launches.  The backward sweep takes the row pairs of the two groups in one
    launch (k_bwd_pair2): the same bits as one pair per launch (BA_SPARSE_BWD2=0).  1 095 cameras: 77 tile rows -- an ODD
    number, the sweep starts with a single row and the forward pairs stay aligned (the Final-13682 shape has 963)."""
    p = ba.synthetic.make_problem(ncams, 6000, 30000, seed=35, locality=0.1)
    if shuffle:
        p, _ = ba.synthetic.shuffle_cameras(p, seed=8)
    arrays = ba.synthetic.as_arrays(p)

    def step(facto_type=None):
        m = ba.BALNLPModel(arrays=arrays)
        d1, h1, _ = ba.lm_step(m, p["x0"], 10.0, facto_type=facto_type)
        d2, h2, _ = ba.lm_step(m, p["x0"], 10.0, facto_type=facto_type)
        pat, (perm, name) = ba.schur_pattern(m), ba.schur_ordering_used(m)
        m.close()
        return d1, d2, h1, h2, pat, perm, name

    a1, a2, ha1, ha2, pat, perm, name = step()
    b1, _, hb1, _, _, _, _ = _env("BA_SPARSE_TWO_RUNS", "0", step)
    f1, f2, _, _, _, _, _ = step(np.float32)
    # the look-ahead inside the two-run phase (lead strips of both runs, both rests on a part of the chip beside the next step's
    # chain), forced on however short the rests are, against the two runs strictly in order: the same bits
    l1, l2, _, _, _, _, _ = _env("BA_SPARSE_LOOKAHEAD_MIN", "1", step)
    n1, _, _, _, _, _, _ = _env("BA_SPARSE_LOOKAHEAD", "0", step)
    l32, _, _, _, _, _, _ = _env("BA_SPARSE_LOOKAHEAD_MIN", "1", lambda: step(np.float32))
    n32, _, _, _, _, _, _ = _env("BA_SPARSE_LOOKAHEAD", "0", lambda: step(np.float32))
    w1, _, _, _, _, _, _ = _env("BA_SPARSE_BWD2", "0", step)
    w32, _, _, _, _, _, _ = _env("BA_SPARSE_BWD2", "0", lambda: step(np.float32))
    for tag, (x, y) in {"backward sweep, two groups per launch vs one pair per launch": (a1, w1), "Float32: the same": (f1, w32)}.items():
        rep = bits_report(x, y, tag)
        assert not rep, rep
    for tag, (x, y) in {"two runs with look-ahead, twice": (l1, l2), "two runs, look-ahead vs in order": (l1, n1),
                        "two runs, look-ahead forced vs default threshold": (l1, a1), "Float32: look-ahead vs in order": (l32, n32)}.items():
        rep = bits_report(x, y, tag)
        assert not rep, rep
    assert name == "two-ended" and pat[2], (name, pat)
    rc, d_ref, dr_ref, _ = orc.lm_step(p["ncams"], p["npnts"], p["cam_idx1"], p["pnt_idx1"], p["pt2d"], p["x0"], 10.0, cam_perm1=perm)
    assert rc == 0
    e_orc, e_one, e32 = rel_err(a1, d_ref), rel_err(a1, b1), rel_err(f1, d_ref)
    print(f"two chains ({'shuffled' if shuffle else 'as generated'}): vs oracle {e_orc:.2e}, vs one chain {e_one:.2e}, Float32 vs oracle {e32:.2e}; pattern {pat}")
    assert e_orc <= 1e-9 and e_one <= 1e-11 and e32 <= 5e-3
    assert abs(ha1 - 0.5 * float(dr_ref @ dr_ref)) <= 1e-10 * ha1 and ha1 == ha2
    for tag, (x, y) in {"Float64 step twice": (a1, a2), "Float32-factor step twice": (f1, f2)}.items():
        rep = bits_report(x, y, tag)
        assert not rep, rep

    def run(graph):
        m = ba.BALNLPModel(arrays=arrays)
        st = _env("BA_LM_GRAPH", graph, lambda: ba.Levenberg_Marquardt(ba.FeasibilityResidual(m), "LDL", "AMD", "None", False, ite_max=4))
        m.close()
        return st

    s_graph, s_plain = run(None), run("0")
    assert s_graph.iter == s_plain.iter and s_graph.log == s_plain.log
    rep = bits_report(s_graph.solution, s_plain.solution, "solution of a 5-iteration run, recorded graphs vs plain launches")
    assert not rep, rep


def test_lm_solve_with_the_iterate_on_the_device_same_bits(ba, gpu_ok):
    """ba_lm_solve_dev (the iterate resident in device memory: what bench.py times) against ba_lm_solve (host vectors, the
    reference's boundary): same iterations, status, log rows and the same bits in the solution."""
    import ctypes
    p = ba.synthetic.make_problem(40, 900, 4200, seed=17)
    m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
    fr = ba.FeasibilityResidual(m)
    a = ba.Levenberg_Marquardt(fr, "LDL", "AMD", "None", False)
    L = ba._lib.lib()
    nbytes = 8 * m.meta.nvar
    d_x = ctypes.c_void_p()
    ba._lib.check(L.ba_dev_malloc(m.handle, nbytes, ctypes.byref(d_x)))
    try:
        x0 = np.ascontiguousarray(p["x0"], dtype=np.float64)
        ba._lib.check(L.ba_memcpy_h2d(m.handle, d_x, ba._lib.ptr(x0), nbytes))
        b = ba.Levenberg_Marquardt(fr, "LDL", "AMD", "None", False, x_device_ptr=d_x.value)
        out = np.empty(m.meta.nvar)
        ba._lib.check(L.ba_memcpy_d2h(m.handle, ba._lib.ptr(out), d_x, nbytes))
    finally:
        ba._lib.check(L.ba_dev_free(m.handle, d_x))
        m.close()
    assert b.solution is None and a.iter == b.iter and a.status == b.status and a.log == b.log and a.objective == b.objective
    rep = bits_report(a.solution, out, "solution: ba_lm_solve vs ba_lm_solve_dev")
    assert not rep, rep


def _env(key, value, fn):
    old = os.environ.get(key)
    if value is None:
        os.environ.pop(key, None)
    else:
        os.environ[key] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = old


@pytest.mark.parametrize("shuffle", [False, True])
def test_block_sparse_lookahead_same_bits_as_in_order(ba, gpu_ok, shuffle):
    """The list schedule of the block-sparse reduced camera system with its look-ahead (the rest of a pair's trailing update
    on a second stream beside the next pair's panel chain, joined by events; dense_ldl_factor_sparse) against the strictly
    in-order form (BA_SPARSE_LOOKAHEAD=0): every tile receives the same updates in the same order, so the steps must be the
    SAME BITS -- plain launches and recorded graphs, Float64 and Float32, twice in a row.  Problem: 700 cameras (n = 6300,
    50 tile rows), cameras of a point within 20 % of the cameras (rests long enough to fork), as generated and with the
    cameras renumbered at random (then the ordering's sequence gives the pattern)."""
    p = ba.synthetic.make_problem(700, 7000, 40000, seed=33, locality=0.2)
    if shuffle:
        p, _ = ba.synthetic.shuffle_cameras(p, seed=4)
    arrays = ba.synthetic.as_arrays(p)

    def step(facto_type=None):
        m = ba.BALNLPModel(arrays=arrays)
        d1, h1, _ = ba.lm_step(m, p["x0"], 10.0, facto_type=facto_type)
        d2, h2, _ = ba.lm_step(m, p["x0"], 10.0, facto_type=facto_type)
        pat = ba.schur_pattern(m)
        m.close()
        return d1, d2, h1, h2, pat

    def run(graph=None):
        m = ba.BALNLPModel(arrays=arrays)
        st = _env("BA_LM_GRAPH", graph, lambda: ba.Levenberg_Marquardt(ba.FeasibilityResidual(m), "LDL", "AMD", "None", False, ite_max=5))
        m.close()
        return st

    os.environ["BA_SPARSE_LOOKAHEAD_MIN"] = "1"  # fork however short the rest is (default: 256 tiles; restored below)
    try:
        _lookahead_checks(ba, step, run)
    finally:
        os.environ.pop("BA_SPARSE_LOOKAHEAD_MIN", None)


def _lookahead_checks(ba, step, run):
    for ft in (None, np.float32):
        a1, a2, ha1, ha2, pat = _env("BA_SPARSE_S", "1", lambda: step(ft))
        b1, b2, hb1, hb2, _ = _env("BA_SPARSE_LOOKAHEAD", "0", lambda: _env("BA_SPARSE_S", "1", lambda: step(ft)))
        assert pat[2] and np.all(np.isfinite(a1))
        tag = "Float32" if ft is not None else "Float64"
        for name, (x, y) in {"look-ahead twice": (a1, a2), "in order twice": (b1, b2), "look-ahead vs in order": (a1, b1)}.items():
            rep = bits_report(x, y, f"{tag} block-sparse step, {name}")
            assert not rep, rep
        assert ha1 == ha2 == hb1 == hb2
    # complete runs: recorded graphs (the forks and joins become graph edges) and plain launches, against the in-order schedule
    s_graph = _env("BA_SPARSE_S", "1", lambda: run(None))
    s_plain = _env("BA_SPARSE_S", "1", lambda: run("0"))
    s_inord = _env("BA_SPARSE_LOOKAHEAD", "0", lambda: _env("BA_SPARSE_S", "1", lambda: run(None)))
    assert s_graph.iter == s_plain.iter == s_inord.iter and s_graph.log == s_plain.log == s_inord.log
    for name, (x, y) in {"graph vs plain launches": (s_graph.solution, s_plain.solution), "look-ahead vs in order": (s_graph.solution, s_inord.solution)}.items():
        rep = bits_report(x, y, f"solution of a 6-iteration run, {name}")
        assert not rep, rep


@pytest.mark.parametrize("variant", ["lm.jl", "LevenbergMarquardt.jl", "lm.jl linesearch"])
def test_prefetched_trial_step_same_rows_as_two_submissions(ba, gpu_ok, variant):
    """Recorded launch sequences (small problems): after an accepted step the refresh of J and the NEXT trial step are
    submitted together and waited for once (ba_lm.hip, accept_refresh_and_trial).  With BA_LM_PREFETCH=0 every sequence
    is submitted and waited for on its own, as before; with BA_LM_GRAPH=0 nothing is recorded at all (plain launches, the
    scalars fetched by a launch of their own).  All three forms must produce the same history rows, the same counters (a
    prefetched step that the stopping tests then drop is not counted) and the same bits in x."""
    p = ba.synthetic.make_named("ladybug-49")
    arrays = ba.synthetic.as_arrays(p)
    args = {"lm.jl": ("None", False), "LevenbergMarquardt.jl": ("None",), "lm.jl linesearch": ("J", True)}[variant]
    forms = [("prefetched", {}), ("separate submissions", {"BA_LM_PREFETCH": "0"}), ("plain launches", {"BA_LM_GRAPH": "0"})]
    out = []
    for _, env in forms:
        os.environ.update(env)
        try:
            m = ba.BALNLPModel(arrays=arrays)
            st = ba.Levenberg_Marquardt(ba.FeasibilityResidual(m), "LDL", "AMD", *args, ite_max=25, atol=0.0, rtol=0.0,
                                        oatol=0.0, ortol=0.0)  # (no early first-order / objective stop: a run of 25 iterations)
            m.close()
        finally:
            for k in env:
                del os.environ[k]
        out.append(st)
    a = out[0]
    assert a.n_accepted >= 5, f"{variant}: only {a.n_accepted} accepted steps -- the prefetch never ran"
    la = np.array(a.log, dtype=np.float64)  # (a NaN rho equals a NaN rho below)
    for (name, _), b in zip(forms[1:], out[1:]):
        ca = (a.status, a.iter, a.n_accepted, a.n_rejected, a.n_factor, a.n_jacobian, a.n_residual)
        cb = (b.status, b.iter, b.n_accepted, b.n_rejected, b.n_factor, b.n_jacobian, b.n_residual)
        assert ca == cb, f"{variant}: counters differ: prefetched {ca} vs {name} {cb}"
        lb = np.array(b.log, dtype=np.float64)
        same = la.shape == lb.shape and np.array_equal(la, lb, equal_nan=True)
        assert same, f"{variant}: history rows of prefetched vs {name} differ ({la.shape} vs {lb.shape}), first at row " \
                     f"{next((i for i in range(min(len(la), len(lb))) if not np.array_equal(la[i], lb[i], equal_nan=True)), min(len(la), len(lb)))}"
        rep = bits_report(a.solution, b.solution, f"{variant}: solution, prefetched trial steps vs {name}")
        assert not rep, rep

# ---- several ranks in one process over the stream-ordered loopback transport ------------------------------------------------
@pytest.fixture(scope="module")
def loopback(gpu_ok):
    assert os.path.exists(LOOPBACK), f"{LOOPBACK} is missing: __graft_entry__.build() compiles it"
    L = C.CDLL(LOOPBACK)
    L.ba_loopback_create.restype = C.c_void_p
    L.ba_loopback_create.argtypes = [C.c_int, C.c_size_t]
    L.ba_loopback_destroy.argtypes = [C.c_void_p]
    L.ba_loopback_rank.restype = C.c_void_p
    L.ba_loopback_rank.argtypes = [C.c_void_p, C.c_int]
    L.ba_loopback_ops.restype = C.c_long
    L.ba_loopback_ops.argtypes = [C.c_void_p]
    return L


class _Ranks:
    """`world` shards of one problem as handles in this process, attached to one loopback communicator."""

    def __init__(self, ba, L, prob, world, stage_mb=64):
        self.ba, self.L, self.world, self.prob = ba, L, world, prob
        arrays = ba.synthetic.as_arrays(prob)
        self.loop = L.ba_loopback_create(world, stage_mb << 20)
        assert self.loop, "loopback communicator could not be created"
        hook = C.cast(L.ba_loopback_hook, ba._lib.COMM_CB)
        self.shards, self.models = [], []
        for r in range(world):
            local, info = ba.parallel.shard_problem(arrays, r, world)
            m = ba.BALNLPModel(arrays=local, device=0)
            ba._lib.check(ba._lib.lib().ba_lm_set_comm_hook(m.handle, r, world, hook, L.ba_loopback_rank(self.loop, r)))
            self.shards.append((local, info))
            self.models.append(m)

    def step(self, lam, **kw):
        """one sharded LM step, every rank on its own host thread -> (global delta from rank 0's cameras, per-rank camera
        parts, model value)"""
        out, err = [None] * self.world, [None] * self.world

        def run(r):
            try:
                out[r] = self.ba.lm_step(self.models[r], self.shards[r][0][3], lam, **kw)
            except Exception as e:  # noqa: BLE001 -- reported below with the rank
                err[r] = e

        ts = [threading.Thread(target=run, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        bad = [(r, e) for r, e in enumerate(err) if e is not None]
        assert not bad, f"rank(s) failed: {bad}"
        ncams, npnts = self.prob["ncams"], self.prob["npnts"]
        delta = np.zeros(3 * npnts + 9 * ncams)
        cams = []
        for r in range(self.world):
            pb, pe = self.shards[r][1]["point_range"]
            d = out[r][0]
            delta[3 * pb:3 * pe] = d[:3 * (pe - pb)]
            cams.append(d[3 * (pe - pb):].copy())
        delta[3 * npnts:] = cams[0]
        return delta, cams, out[0][1]

    def close(self):
        for m in self.models:
            m.close()
        self.L.ba_loopback_destroy(self.loop)


def _set_lookahead(on):
    if on:
        os.environ.pop("BA_DIST_LOOKAHEAD", None)
    else:
        os.environ["BA_DIST_LOOKAHEAD"] = "0"


@pytest.mark.parametrize("world", [2, 3])
def test_loopback_sharded_step_equals_one_rank(ba, loopback, world):
    """Observations sharded by point over 2 / 3 ranks of ONE process, reduce of the reduced camera matrix onto the owners,
    distributed factorisation with look-ahead on asynchronous streams: the step of the unsharded problem to rounding
    (Float64) / Float32 level (facto_type = Float32), the camera part bit-identical on every rank."""
    prob = ba.synthetic.make_problem(200, 1500, 9000, seed=11)  # n = 1800: 15 tile rows, 8 tile column pairs
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
    d_ref, half_ref, _ = ba.lm_step(ref, prob["x0"], 10.0)
    d32_ref, _, _ = ba.lm_step(ref, prob["x0"], 10.0, facto_type=np.float32)
    ref.close()
    _set_lookahead(True)
    R = _Ranks(ba, loopback, prob, world)
    try:
        d, cams, half = R.step(10.0)
        e = rel_err(d, d_ref)
        assert e <= 1e-9, f"{world} ranks, Float64: |delta - delta_one_rank| / |delta_one_rank| = {e:.3e} (limit 1e-9)"
        assert abs(half - half_ref) <= 1e-10 * half_ref, f"{world} ranks: model value {half!r} vs {half_ref!r}"
        for r in range(1, world):
            rep = bits_report(cams[0], cams[r], f"camera step of rank 0 vs rank {r} (replicated solve)")
            assert not rep, rep
        d32, cams32, _ = R.step(10.0, facto_type=np.float32)
        e32 = rel_err(d32, d32_ref)
        assert e32 <= 5e-3, f"{world} ranks, Float32 factor: relative step difference {e32:.3e} (limit 5e-3)"
        for r in range(1, world):
            rep = bits_report(cams32[0], cams32[r], f"Float32-factor camera step of rank 0 vs rank {r}")
            assert not rep, rep
        dp, _, _ = R.step(10.0, pcg=(1e-12, 5000))[:3]
        ep = rel_err(dp, d_ref)
        assert ep <= 1e-8, f"{world} ranks, facto = :PCG: relative step difference {ep:.3e} (limit 1e-8)"
        assert loopback.ba_loopback_ops(R.loop) > 0
        # per-rank ownership of S: a rank holds its own tile columns (about 1 / world of the triangle: whole column pairs, so
        # up to one pair more than the even share) plus a staging chunk of at most about half of that
        for r, m in enumerate(R.models):
            full, held, staging = ba.schur_memory(m)
            nt = 15
            slack = 2 * nt  # one tile column pair
            assert full == nt * (nt + 1) // 2
            assert held <= full / world + slack, f"rank {r}: holds {held} of {full} tiles ({world} ranks)"
            assert staging <= 0.5 * (full / world + slack) + nt, f"rank {r}: staging {staging} tiles, own share {held}"
            assert held + staging <= 1.5 * full / world + 2 * slack, f"rank {r}: {held} + {staging} tiles against 1.5 x {full} / {world}"
    finally:
        R.close()


@pytest.mark.parametrize("ncams,npnts,nobs", [(200, 1500, 9000), (640, 5000, 36000)])
def test_loopback_lookahead_same_bits_as_alternating(ba, loopback, ncams, npnts, nobs):
    """Distributed factorisation on 3 ranks with asynchronous streams: the look-ahead schedule (owner of the next pair
    updates its leading columns first, chain + panel hand-over on the transfer stream beside the rest of the update)
    against the strictly alternating one (BA_DIST_LOOKAHEAD=0) -- the same products in the same order, so the SAME BITS;
    and the look-ahead run twice gives the same bits (n = 1800: 8 tile column pairs; n = 5760: 23 pairs, updates long
    enough to overlap the transfers for real)."""
    prob = ba.synthetic.make_problem(ncams, npnts, nobs, seed=11)
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
    d_ref, _, _ = ba.lm_step(ref, prob["x0"], 10.0)
    ref.close()
    R = _Ranks(ba, loopback, prob, 3, stage_mb=256)
    try:
        runs = {}
        for tag, on in (("look-ahead, run 1", True), ("alternating", False), ("look-ahead, run 2", True)):
            _set_lookahead(on)
            d, cams, _ = R.step(10.0)
            e = rel_err(d, d_ref)
            assert e <= 1e-9, f"{tag}: relative difference to the one-rank step {e:.3e} (limit 1e-9)"
            for r in (1, 2):
                rep = bits_report(cams[0], cams[r], f"{tag}: camera step of rank 0 vs rank {r}")
                assert not rep, rep
            runs[tag] = cams[0]
        print({k: digest(v) for k, v in runs.items()})
        for tag in ("alternating", "look-ahead, run 2"):
            rep = bits_report(runs["look-ahead, run 1"], runs[tag], f"camera step, look-ahead run 1 vs {tag}")
            assert not rep, rep
    finally:
        _set_lookahead(True)
        R.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("irregular", [False, True])
def test_loopback_block_sparse_on_several_ranks(ba, loopback, world, irregular):
    """The list schedule of the block-sparse reduced camera system on 2 / 3 ranks of one process (per-rank ownership of the
    PATTERN's tile columns, chunked assembly over pattern tiles, panel broadcast of pattern rows, list-driven updates and
    sweeps): the one-rank step to 1e-9 (Float64) / 5e-3 (Float32 factorisation), the camera part bit-identical on every
    rank, look-ahead = alternating bit for bit, and a rank holds at most 1.5 x its share of the PATTERN.  Problems: 520
    cameras (n = 4680, 37 tile rows), cameras of a point within 14 % of the cameras -- a band -- and the same with the
    cameras renumbered in whole groups (an irregular pattern: row lists with gaps, broadcasts in several runs); the pattern
    of a rank's own observations differs from rank to rank, the handles agree on the pattern of the sum."""
    p0 = ba.synthetic.make_problem(520, 5200, 26000, seed=12, locality=0.08 if irregular else 0.14)
    if irregular:  # whole groups of 14 cameras (a tile's worth) shuffled inside windows of six groups
        rng = np.random.default_rng(12)
        ng = 520 // 14
        gperm = np.arange(ng)
        for g0 in range(0, ng, 6):
            gperm[g0:g0 + 6] = rng.permutation(gperm[g0:g0 + 6])
        sigma = np.arange(520)
        for g in range(ng):
            sigma[g * 14:(g + 1) * 14] = gperm[g] * 14 + np.arange(14)
        p0, _ = ba.synthetic.shuffle_cameras(p0, sigma=sigma)
    prob = p0
    os.environ["BA_SPARSE_S"] = "1"
    os.environ["BA_CAM_ORDER"] = "natural"  # (the numbering is the test's own: irregular on purpose)
    try:
        ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
        d_ref, half_ref, _ = ba.lm_step(ref, prob["x0"], 10.0)
        d32_ref, _, _ = ba.lm_step(ref, prob["x0"], 10.0, facto_type=np.float32)
        tf, ff, sparse_one = ba.schur_pattern(ref)
        full_one, held_one, _ = ba.schur_memory(ref)
        ref.close()
        assert sparse_one and held_one < full_one
        R = _Ranks(ba, loopback, prob, world, stage_mb=256)
        try:
            runs = {}
            for tag, on in (("look-ahead", True), ("alternating", False)):
                _set_lookahead(on)
                d, cams, half = R.step(10.0)
                e = rel_err(d, d_ref)
                assert e <= 1e-9, f"{world} ranks, {tag}: |delta - delta_one_rank| / |delta_one_rank| = {e:.3e} (limit 1e-9)"
                assert abs(half - half_ref) <= 1e-10 * half_ref
                for r in range(1, world):
                    rep = bits_report(cams[0], cams[r], f"{tag}: camera step of rank 0 vs rank {r}")
                    assert not rep, rep
                runs[tag] = cams[0]
            rep = bits_report(runs["look-ahead"], runs["alternating"], "camera step, look-ahead vs alternating")
            assert not rep, rep
            _set_lookahead(True)
            d32, cams32, _ = R.step(10.0, facto_type=np.float32)
            e32 = rel_err(d32, d32_ref)
            assert e32 <= 5e-3, f"{world} ranks, Float32 factor: relative step difference {e32:.3e} (limit 5e-3)"
            held_sum = 0
            for r, m in enumerate(R.models):
                t, f, sp = ba.schur_pattern(m)
                full, held, staging = ba.schur_memory(m)
                assert sp and abs(t - tf) < 1e-12 and abs(f - ff) < 1e-12, f"rank {r}: pattern {t, f, sp} vs one rank {tf, ff}"
                slack = 2 * 37  # one tile column pair
                assert held <= held_one / world + slack, f"rank {r}: holds {held} tiles, pattern {held_one}, {world} ranks"
                assert held + staging <= 1.5 * held_one / world + 2 * slack, f"rank {r}: {held} + {staging} tiles against 1.5 x {held_one} / {world}"
                held_sum += held
            assert held_sum == held_one, f"the ranks hold {held_sum} tiles together, the pattern has {held_one}"
            print(f"{world} ranks, {'irregular' if irregular else 'band'}: pattern {held_one} of {full_one} tiles, per rank "
                  f"{[ba.schur_memory(m)[1:] for m in R.models]}")
        finally:
            _set_lookahead(True)
            R.close()
    finally:
        os.environ.pop("BA_SPARSE_S", None)
        os.environ.pop("BA_CAM_ORDER", None)


@pytest.mark.parametrize("world,ncams,npnts,nobs", [(2, 200, 1500, 9000), (3, 640, 5000, 36000)])
def test_loopback_reduce_scatter_assembly_equals_reduce_onto_owner(ba, loopback, world, ncams, npnts, nobs):
    """The chunked assembly of the reduced camera matrix in its reduce-scatter form (default: a chunk = one slice of every
    owner's tile columns, one in-place reduce-scatter, two staging buffers so that a chunk travels on the transfer stream
    while the next is assembled) against round 3's form (BA_ASSEMBLY=reduce: a chunk = a range of ONE owner's columns,
    reduced onto it).  The loopback transport sums in rank order in both, so the steps must be the SAME BITS (Float64 and
    Float32 factorisation); the per-operation counters show one reduce-scatter per chunk and no reduce; a rank's staging
    stays within half its share of S (small problems: one buffer) and the second run of the same handles gives the same bits."""
    prob = ba.synthetic.make_problem(ncams, npnts, nobs, seed=11)
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
    d_ref, _, _ = ba.lm_step(ref, prob["x0"], 10.0)
    ref.close()
    out = {}
    for form in ("reduce-scatter", "reduce"):
        if form == "reduce":
            os.environ["BA_ASSEMBLY"] = "reduce"
        try:
            R = _Ranks(ba, loopback, prob, world, stage_mb=512)
            try:
                d, cams, _ = R.step(10.0)
                d2, cams2, _ = R.step(10.0)
                d32, cams32, _ = R.step(10.0, facto_type=np.float32)
                e = rel_err(d, d_ref)
                assert e <= 1e-9, f"{form}: relative difference to the one-rank step {e:.3e}"
                rep = bits_report(cams[0], cams2[0], f"{form}: first vs second step on the same handles")
                assert not rep, rep
                for r in range(1, world):
                    rep = bits_report(cams[0], cams[r], f"{form}: camera step of rank 0 vs rank {r}")
                    assert not rep, rep
                calls = (C.c_int64 * ba._lib.COMM_OPS)()
                nbytes = (C.c_int64 * ba._lib.COMM_OPS)()
                ba._lib.check(ba._lib.lib().ba_comm_stats_ops(R.models[0].handle, calls, nbytes))
                ops = {ba._lib.COMM_OP_NAMES[q]: (int(calls[q]), int(nbytes[q])) for q in range(ba._lib.COMM_OPS)}
                mem = [ba.schur_memory(m) for m in R.models]
                out[form] = (cams[0], cams32[0], ops, mem)
            finally:
                R.close()
        finally:
            os.environ.pop("BA_ASSEMBLY", None)
    for tag, idx in (("Float64", 0), ("Float32 factorisation", 1)):
        rep = bits_report(out["reduce-scatter"][idx], out["reduce"][idx], f"{tag} camera step, reduce-scatter vs reduce-onto-owner assembly")
        assert not rep, rep
    ops_rs, ops_red = out["reduce-scatter"][2], out["reduce"][2]
    print("reduce-scatter form:", {k: v for k, v in ops_rs.items() if v[0]}, "memory", out["reduce-scatter"][3])
    print("reduce form:        ", {k: v for k, v in ops_red.items() if v[0]}, "memory", out["reduce"][3])
    assert ops_rs["reduce_scatter_f64"][0] > 0 and ops_rs["reduce_scatter_f32"][0] > 0 and ops_rs["reduce_f64"][0] == 0 and ops_rs["reduce_f32"][0] == 0
    assert ops_red["reduce_scatter_f64"][0] == 0 and ops_red["reduce_f64"][0] > 0
    # all staging within half the largest share plus two tile columns (slices are whole tile columns: on a matrix this small a
    # column is a quarter of a share)
    nt = int(np.sqrt(2 * out["reduce-scatter"][3][0][0]))
    biggest = max(h for _, h, _ in out["reduce-scatter"][3])
    for (full, held, staging) in out["reduce-scatter"][3]:
        assert staging <= 0.5 * biggest + 2 * nt + 2, f"staging {staging} tiles, largest share {biggest}, {nt} tile rows"


@pytest.mark.parametrize("scene", ["band", "plane"])
def test_loopback_camera_ordering_on_several_ranks(ba, loopback, scene):
    """A randomly numbered block-banded problem on 3 ranks: every rank sees only the camera pairs ITS points connect; the
    camera graph is summed over the ranks before it is ordered, so every handle arrives at the sequence (and the pattern)
    of the one-rank handle, takes the list schedule, and the step is the one-rank step.  scene = plane: cameras standing in
    the plane (a two-dimensional geometric camera graph, irregular row lists) instead of a band."""
    if scene == "band":
        p0 = ba.synthetic.make_problem(520, 5200, 26000, seed=13, locality=0.12)
        prob, _ = ba.synthetic.shuffle_cameras(p0, seed=6)
    else:
        prob = ba.synthetic.make_problem(520, 5200, 26000, seed=14, plane_radius=0.11)
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
    d_ref, half_ref, _ = ba.lm_step(ref, prob["x0"], 10.0)
    pat_ref, (perm_ref, name_ref) = ba.schur_pattern(ref), ba.schur_ordering_used(ref)
    ref.close()
    assert pat_ref[2] and name_ref != "natural"
    R = _Ranks(ba, loopback, prob, 3, stage_mb=256)
    try:
        d, cams, half = R.step(10.0)
        e = rel_err(d, d_ref)
        assert e <= 1e-9, f"3 ranks with a camera ordering: relative difference to the one-rank step {e:.3e}"
        for r, m in enumerate(R.models):
            perm, name = ba.schur_ordering_used(m)
            assert np.array_equal(perm, perm_ref) and name == name_ref, f"rank {r}: sequence '{name}' differs from the one-rank handle's '{name_ref}'"
            assert ba.schur_pattern(m) == pat_ref, f"rank {r}: pattern {ba.schur_pattern(m)} vs {pat_ref}"
            if r:
                rep = bits_report(cams[0], cams[r], f"camera step of rank 0 vs rank {r}")
                assert not rep, rep
    finally:
        R.close()


def test_loopback_lm_runs_equal_one_rank(ba, loopback):
    """Complete LM runs on 3 ranks of one process over the loopback transport (per-rank ownership of S, distributed
    factorisation and backward sweep, chunked assembly): lm.jl with and without column scaling -- the scaling uses the
    all-reduced diagonal of J'J and touches only the tiles a rank owns -- give the one-rank run's iterations, status and
    objective."""
    prob = ba.synthetic.make_problem(70, 500, 2400, seed=8)  # n = 630: 5 tile rows, pairs owned 0, 1, 2
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob))
    want = {}
    for norm in ("None", "J", "A"):
        want[norm] = ba.Levenberg_Marquardt(ba.FeasibilityResidual(ref), "LDL", "AMD", norm, False)
    ref.close()
    _set_lookahead(True)
    R = _Ranks(ba, loopback, prob, 3)
    try:
        for norm in ("None", "J", "A"):
            out, err = [None] * 3, [None] * 3

            def run(r, norm=norm):
                try:
                    out[r] = ba.Levenberg_Marquardt(ba.FeasibilityResidual(R.models[r]), "LDL", "AMD", norm, False)
                except Exception as e:  # noqa: BLE001
                    err[r] = e

            ts = [threading.Thread(target=run, args=(r,)) for r in range(3)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            assert not any(err), f"normalize {norm}: {err}"
            w = want[norm]
            for r in range(3):
                st = out[r]
                assert st.iter == w.iter and st.status == w.status, f"normalize {norm}, rank {r}: {st.iter} {st.status} vs {w.iter} {w.status}"
                assert abs(st.objective - w.objective) <= 1e-9 * w.objective, f"normalize {norm}, rank {r}: {st.objective!r} vs {w.objective!r}"
            cam = [out[r].solution[-9 * prob["ncams"]:] for r in range(3)]
            for r in (1, 2):
                rep = bits_report(cam[0], cam[r], f"normalize {norm}: cameras of rank 0 vs rank {r}")
                assert not rep, rep
    finally:
        R.close()
