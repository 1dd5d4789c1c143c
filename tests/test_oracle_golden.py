"""The oracle (oracle/ba_oracle.c) against every golden vector / known answer the reference's own tests hold for
this path (test/runtests.jl) and against residuals produced by the reference's Python restatement
(src/SolverScipy.py `fun`, committed under tests/golden/ by tests/golden/make_golden.py).  CPU only."""
import numpy as np


def test_runtests_residual_fixture_bit_exact(orc, fixture_runtests):
    f = fixture_runtests  # test/runtests.jl:15-27: norm(true_residuals - r) == 0
    r = orc.residuals(f["cam_idx"], f["pnt_idx"], f["x"], f["pt2d"], int(f["npnts"]))
    assert np.linalg.norm(f["true_residuals"] - r) == 0


def test_runtests_known_answers_bit_exact(orc, fixture_runtests):
    f = fixture_runtests
    # test/runtests.jl:6  Rodrigues_rotation([1,1,1],[2.5,-0.3,1.0])  (== P1 with t = 0)
    rot = orc.P1(f["rodrigues_r"], np.zeros(3), f["rodrigues_x"])
    assert np.array_equal(rot, f["rodrigues_out"])
    # test/runtests.jl:8  projection(x,y,z, rx,ry,rz, tx,ty,tz, f,k1,k2) == [-7 -7]
    a = f["projection_args"]
    cam = np.array([a[3], a[4], a[5], a[6], a[7], a[8], a[10], a[11], a[9]])  # (r, t, k1, k2, f)
    assert np.array_equal(orc.projection(a[:3], cam), f["projection_out"])
    # test/runtests.jl:7  scaling_factor([1 1], 1, 1) == 7: the origin rotates to exactly 0, so P1 = t = (-1,-1,1),
    # P2 = (1,1) and projection = f * scaling * P2 = (7,7) for f = 1
    cam = np.array([1.0, 1.0, 1.0, -1.0, -1.0, 1.0, 1.0, 1.0, 1.0])
    out = orc.projection(np.zeros(3), cam)
    assert np.array_equal(out, np.array([7.0, 7.0]))


def test_residuals_match_reference_python_bit_exact(orc, golden_scipy):
    g = golden_scipy
    for tag in ("x0", "xtrue"):
        x = g["x0"] if tag == "x0" else g["x_true"]
        r = orc.residuals(g["cam_idx1"], g["pnt_idx1"], x, g["pt2d"], int(g["npnts"]))
        assert np.array_equal(r, g["res_" + tag])


def test_f32_twin_close_to_f64(orc, golden_scipy):
    g = golden_scipy
    r64 = orc.residuals(g["cam_idx1"], g["pnt_idx1"], g["x0"], g["pt2d"], int(g["npnts"]))
    r32 = orc.residuals(g["cam_idx1"], g["pnt_idx1"], g["x0"].astype(np.float32), g["pt2d"].astype(np.float32),
                        int(g["npnts"]))
    assert r32.dtype == np.float32
    assert np.max(np.abs(r32 - r64)) < 0.5  # pixels; Float32 carries ~1e-4 relative on |proj| ~ 1e3


def test_jac_structure_closed_form(orc, small_prob):
    p = small_prob
    rows, cols = orc.jac_structure(p["cam_idx1"], p["pnt_idx1"], p["npnts"])
    k = np.arange(p["nobs"])
    exp_rows = np.repeat(np.stack([2 * k + 1, 2 * k + 2], 1), 12, axis=1).reshape(-1)
    pc = 3 * (p["pnt_idx1"] - 1)[:, None] + 1 + np.arange(3)[None, :]
    cc = 3 * p["npnts"] + 9 * (p["cam_idx1"] - 1)[:, None] + 1 + np.arange(9)[None, :]
    one = np.concatenate([pc, cc], 1)
    exp_cols = np.concatenate([one, one], 1).reshape(-1)
    assert np.array_equal(rows, exp_rows) and np.array_equal(cols, exp_cols)
    assert rows.dtype == np.int64 and rows.min() == 1 and cols.max() == 3 * p["npnts"] + 9 * p["ncams"]


def _complex_step_block(X, C):
    """Independent derivative of the pinned projection (complex step, h = 1e-30): 2x12 in column order
    [X, r, t, k1, k2, f]."""
    def proj(X, C):
        r, t, k1, k2, f = C[:3], C[3:6], C[6], C[7], C[8]
        th = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
        k = r / th
        c, s = np.cos(th), np.sin(th)
        kx = np.array([k[1] * X[2] - k[2] * X[1], k[2] * X[0] - k[0] * X[2], k[0] * X[1] - k[1] * X[0]])
        d = k[0] * X[0] + k[1] * X[1] + k[2] * X[2]
        P1 = c * X + s * kx + (1 - c) * d * k + t
        P2 = -P1[:2] / P1[2]
        n = P2[0] * P2[0] + P2[1] * P2[1]
        return f * (1.0 + k1 * n + k2 * n * n) * P2
    v = np.concatenate([X, C]).astype(complex)
    J = np.zeros((2, 12))
    for j in range(12):
        w = v.copy()
        w[j] += 1e-30j
        J[:, j] = proj(w[:3], w[3:]).imag / 1e-30
    return J


def test_jac_coord_is_the_true_derivative(orc, small_prob):
    p = small_prob
    vals = orc.jac_coord(p["cam_idx1"], p["pnt_idx1"], p["x0"], p["npnts"]).reshape(-1, 2, 12)
    x = p["x0"]
    rng = np.random.default_rng(0)
    for k in rng.choice(p["nobs"], size=40, replace=False):
        X = x[3 * (p["pnt_idx1"][k] - 1): 3 * p["pnt_idx1"][k]]
        C = x[3 * p["npnts"] + 9 * (p["cam_idx1"][k] - 1): 3 * p["npnts"] + 9 * p["cam_idx1"][k]]
        Jcs = _complex_step_block(X, C)
        assert np.max(np.abs(vals[k] - Jcs)) <= 1e-11 * np.max(np.abs(Jcs))


def test_jac_coord_nan_to_zero(orc):
    # theta = 0 (r = 0) -> every entry NaN in the reference -> 0 (src/BALNLPModels.jl:201)
    cam = np.array([1], dtype=np.int64)
    x = np.array([0.1, 0.2, 0.3, 0, 0, 0, 0, 0, -5.0, 1e-7, 1e-12, 500.0])
    assert np.all(orc.jac_coord(cam, cam, x, 1) == 0)
    assert np.all(np.isnan(orc.residuals(cam, cam, x, np.zeros(2), 1)))
    # P1.z = 0 -> NaN block -> 0; the residual itself is +-Inf/NaN (no guard in the reference)
    x = np.array([0.1, 0.2, 0.0, 0, 0, 0.3, 0, 0, 0.0, 1e-7, 1e-12, 500.0])  # rotation about z keeps P1.z = t_z = 0 exactly
    v = orc.jac_coord(cam, cam, x, 1)
    assert np.all(v == 0)


# ---- test/runtests.jl:29-128: the reference's own checks of lma_aux.jl / qr_aux.jl, replayed on the oracle ---------------
def _dense(colptr, rowval, nzval, m, n):
    A = np.zeros((m, n))
    for j in range(n):
        for p in range(colptr[j], colptr[j + 1]):
            A[rowval[p], j] += nzval[p]
    return A


def _rand_half(rng, k):  # rand(-4.5:4.5, k): the ten half-integers -4.5 ... 4.5
    return rng.integers(0, 10, k) - 4.5


def test_runtests_normalize_qr(orc):
    """runtests.jl:31-63: normalize_qr_a!, normalize_qr_j!, denormalize_qr! on A = [random 7x5 ; sqrt(lambda) I]."""
    L = orc.lib()
    for seed in range(20):
        rng = np.random.default_rng(seed)
        m, n, lam = 7, 5, 1.5
        rows = np.concatenate([rng.integers(1, m + 1, 8), np.arange(m + 1, m + n + 1)]).astype(np.int64)
        cols = np.concatenate([rng.integers(1, n + 1, 8), np.arange(1, n + 1)]).astype(np.int64)
        vals = np.concatenate([_rand_half(rng, 8), np.full(n, np.sqrt(lam))])
        colptr, rowval, nz0 = orc.sparse(rows, cols, vals, m + n, n)
        A0 = _dense(colptr, rowval, nz0, m + n, n)
        nz = nz0.copy()
        cn = np.zeros(n)
        L.orc_normalize_qr_a(colptr, nz, cn, n)
        A = _dense(colptr, rowval, nz, m + n, n)
        for j in range(n):
            assert abs(np.linalg.norm(A[:, j]) - 1) < 1e-10
            assert abs(cn[j] - np.linalg.norm(A0[:, j])) < 1e-10
        L.orc_denormalize_qr(colptr, nz, cn, n)
        assert np.linalg.norm(_dense(colptr, rowval, nz, m + n, n) - A0) < 1e-10
        nz = nz0.copy()
        L.orc_normalize_qr_j(colptr, nz, cn, n)
        A = _dense(colptr, rowval, nz, m + n, n)
        for j in range(n):
            if cn[j] != 0:
                assert abs(np.linalg.norm(A[:m, j]) - 1) < 1e-10
                assert A[m + j, j] - A0[m + j, j] / cn[j] < 1e-10
            assert abs(cn[j] - np.linalg.norm(A0[:m, j])) < 1e-10
        L.orc_denormalize_qr(colptr, nz, cn, n)
        assert np.linalg.norm(_dense(colptr, rowval, nz, m + n, n)[:m] - A0[:m]) < 1e-10


def test_runtests_normalize_ldl(orc):
    """runtests.jl:65-88: normalize_ldl!, denormalize_ldl! on the upper triangle of [[I A12];[A12' -lambda I]]."""
    L = orc.lib()
    for seed in range(20):
        rng = np.random.default_rng(100 + seed)
        m, n, lam = 7, 5, 1.5
        rows = np.concatenate([np.arange(1, m + 1), rng.integers(1, m + 1, 8), np.arange(m + 1, m + n + 1)]).astype(np.int64)
        cols = np.concatenate([np.arange(1, m + 1), m + rng.integers(1, n + 1, 8), np.arange(m + 1, m + n + 1)]).astype(np.int64)
        vals = np.concatenate([np.ones(m), _rand_half(rng, 8), np.full(n, -lam)])
        colptr, rowval, nz0 = orc.sparse(rows, cols, vals, m + n, m + n)
        A0 = _dense(colptr, rowval, nz0, m + n, m + n)
        nz = nz0.copy()
        cn = np.zeros(n)
        L.orc_normalize_ldl(colptr, nz, cn, n, m)
        A = _dense(colptr, rowval, nz, m + n, m + n)
        for j in range(n):
            if cn[j] != 0:
                assert abs(np.linalg.norm(A[:m, m + j]) - 1) < 1e-10
                assert abs(A[m + j, m + j] - A0[m + j, m + j] / cn[j] ** 2) < 1e-10
            assert abs(cn[j] - np.linalg.norm(A0[:m, m + j])) < 1e-10  # (the reference's slice m+1:m+j is a typo for m+j)
        L.orc_denormalize_ldl(colptr, nz, cn, n, m)
        assert np.linalg.norm(_dense(colptr, rowval, nz, m + n, m + n)[:m, m:] - A0[:m, m:]) < 1e-10


def test_runtests_mul_sparse(orc):
    """runtests.jl:91-108: COO spmv (duplicates add up, as sparse() sums them) against A * x."""
    for seed in range(20):
        rng = np.random.default_rng(200 + seed)
        m, n, nnz = 7, 5, 8
        rows = rng.integers(1, m + 1, nnz).astype(np.int64)
        cols = rng.integers(1, n + 1, nnz).astype(np.int64)
        vals = _rand_half(rng, nnz)
        A = np.zeros((m, n))
        np.add.at(A, (rows - 1, cols - 1), vals)
        for _ in range(2):
            x = _rand_half(rng, n)
            assert np.linalg.norm(orc.mul_sparse(rows, cols, vals, x, m) - A @ x) < 1e-10


def test_runtests_least_squares_solve(orc):
    """runtests.jl:111-128 pins the :QR branch (SuiteSparse SPQR, absent here) by x = A \\ b for A = [J; sqrt(lambda) I].
    The build serves :QR and :LDL with one solve of the same normal equations; the oracle's LDL' of the augmented matrix
    [[I J];[J' -lambda I]] with right-hand side [b1; -sqrt(lambda) b2] is that solve, checked against lstsq."""
    for seed in range(20):
        rng = np.random.default_rng(300 + seed)
        m, n, lam = 7, 5, 1.5
        rj = rng.integers(1, m + 1, 8).astype(np.int64)
        cj = rng.integers(1, n + 1, 8).astype(np.int64)
        vj = _rand_half(rng, 8)
        b = _rand_half(rng, m + n)
        J = np.zeros((m, n))
        np.add.at(J, (rj - 1, cj - 1), vj)
        A = np.vstack([J, np.sqrt(lam) * np.eye(n)])
        true_x = np.linalg.lstsq(A, b, rcond=None)[0]
        # upper triangle of K, columns: m identity columns, then [J(:,j); -lambda]
        rows = np.concatenate([np.arange(1, m + 1), rj, np.arange(m + 1, m + n + 1)]).astype(np.int64)
        cols = np.concatenate([np.arange(1, m + 1), m + cj, np.arange(m + 1, m + n + 1)]).astype(np.int64)
        vals = np.concatenate([np.ones(m), vj, np.full(n, -lam)])
        colptr, rowval, nzval = orc.sparse(rows, cols, vals, m + n, m + n)
        rhs = np.concatenate([b[:m], -np.sqrt(lam) * b[m:]])
        rc, x, lnz, D = orc.ldl_solve(colptr, rowval, nzval, np.arange(m + n, dtype=np.int64), rhs)
        assert rc == 0
        assert np.linalg.norm(x[m:] - true_x) < 1e-10


def test_runtests_qr_solve(orc):
    """runtests.jl:111-128 itself: x from the QR of A = [J; sqrt(lambda) I] equals A \\ b to 1e-10.  The oracle's dense
    Householder QR (orc_qr_lstsq) stands in for myqr + solve_qr! (SuiteSparse SPQR is not in the image)."""
    for seed in range(20):
        rng = np.random.default_rng(700 + seed)
        m, n, lam = 7, 5, 1.5
        rj = rng.integers(1, m + 1, 8)
        cj = rng.integers(1, n + 1, 8)
        vj = _rand_half(rng, 8)
        b = _rand_half(rng, m + n)
        J = np.zeros((m, n))
        np.add.at(J, (rj - 1, cj - 1), vj)
        A = np.vstack([J, np.sqrt(lam) * np.eye(n)])
        rc, x = orc.qr_lstsq(A, b)
        assert rc == 0
        assert np.linalg.norm(x - np.linalg.lstsq(A, b, rcond=None)[0]) < 1e-10


def test_oracle_qr_branch_equals_ldl_branch(orc, ba):
    """lm.jl's :QR and :LDL branches solve the same damped least-squares problem: on a small problem the two oracle runs
    must take the same iterations (this is what lets one device solve serve both)."""
    p = ba.synthetic.make_problem(6, 80, 320, seed=9)
    for code in (0, 1, 2):
        out = []
        for facto in ("LDL", "QR"):
            rc, x, st, log = orc.lm_solve(p["ncams"], p["npnts"], p["cam_idx1"], p["pnt_idx1"], p["pt2d"], p["x0"],
                                          variant=1, normalize=code, facto=facto)
            assert rc == 0
            out.append((st.iter, st.status, st.objective, log[:, 1].copy()))
        assert out[0][:2] == out[1][:2]
        assert abs(out[0][2] - out[1][2]) <= 1e-10 * out[0][2]
        assert np.allclose(out[0][3], out[1][3], rtol=1e-8)


def test_oracle_float32_model_loop(ba, orc):
    """orc_lm_solve_f32 (T = Float32, src/lm.jl:15-59,251-337 with eltype(x) = Float32): self-consistency of the restatement --
    with the reference's Float32-experiment tolerances (src/diffprecsions.jl:22) it follows the Float64 loop's accept/reject
    sequence and reaches its minimum to Float32 level; with the eps(Float32)-derived defaults the step test stops it at the
    first accepted step; lambda is a Float32 value before the first accepted step and not afterwards (lm.jl:337)."""
    p = ba.synthetic.make_problem(12, 400, 1800, seed=11)
    a = (p["ncams"], p["npnts"], p["cam_idx1"], p["pnt_idx1"])
    pt32, x32 = p["pt2d"].astype(np.float32), p["x0"].astype(np.float32)
    tol32 = dict(oatol=1e-4, ortol=1e-4, atol=1e-4, rtol=1e-5, satol=1e-6, srtol=1e-7)
    for variant in (1, 0):
        for norm in (0, 1, 2):
            rc, x, st, log = orc.lm_solve_f32(*a, pt32, x32, variant=variant, normalize=norm, **tol32)
            rc64, x64, st64, log64 = orc.lm_solve(*a, p["pt2d"], p["x0"], variant=variant, normalize=norm)
            assert rc == 0 and rc64 == 0 and x.dtype == np.float32
            assert abs(st.objective - st64.objective) <= 1e-5 * st64.objective, (variant, norm, st.objective, st64.objective)
            n = min(len(log), len(log64), 6)
            assert list(log[:n, 7]) == list(log64[:n, 7])
            assert np.allclose(log[:n, 1], log64[:n, 1], rtol=1e-3)  # (a Float32 LDL' of the augmented system: steps to ~1e-4)
    rc, x, st, log = orc.lm_solve_f32(*a, pt32, x32, variant=1)
    assert rc == 0 and orc.STATUS[st.status] == "small_step" and st.iter <= 2
    rc, x, st, log = orc.lm_solve_f32(*a, pt32, x32, variant=1, **tol32)
    lam = log[:, 4]
    assert lam[0] == np.float32(lam[0]) and lam[1] == np.float32(lam[1]) / 1 and np.float64(np.float32(lam[1])) == lam[1]
    assert any(np.float64(np.float32(v)) != v for v in lam[2:]), "lambda must become a Float64 after the first accepted step"
    # the old variant keeps lambda in Float32 throughout (LevenbergMarquardt.jl:269,292)
    rc, x, st, log = orc.lm_solve_f32(*a, pt32, x32, variant=0, **tol32)
    assert all(np.float64(np.float32(v)) == v for v in log[:, 4])
