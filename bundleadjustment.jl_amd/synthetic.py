"""Seeded synthetic BAL-shaped problems (no BAL file is available offline) and a BAL writer.

Generator spec: unit-ball points; rotation vectors axis*angle with angle ~ U(0.05, 1.5)
(never 0: theta = 0 is NaN in the reference, src/BALNLPModels.jl:19-23); camera centres at distance 6 +- 1 looking at
the origin down -z so that P1.z is in (-8, -4) (never 0, :26-31); f ~ U(400, 1800); k1 ~ -U(0,5)e-7; k2 ~ U(0,1)e-12
(magnitudes of the fixture of test/runtests.jl:18); per-point degree >= 2 from a shifted geometric law adjusted so
that the degrees sum to nobs exactly; cameras of a point distinct and uniformly drawn (=> the reduced camera matrix
is dense; `locality` draws them from a window of consecutive cameras instead: block-banded S, see make_problem); observations sorted by point then camera (BAL order, test/runtests.jl:16-17);
pt2d = projection(x_true) + N(0, 0.5^2) px; x0 = x_true with points + N(0, 0.02^2), cameras * (1 + N(0, 1e-3^2)).
"""
import bz2
import os

import numpy as np

# (ncams, npnts, nobs) of the configurations of BASELINE.json
SHAPES = {
    "ladybug-49": (49, 7776, 31843),
    "dubrovnik-356": (356, 226730, 1255268),
    "venice-1778": (1778, 993923, 5001946),
    "final-13682": (13682, 4456117, 28987644),
}
BASE_SEED = 20261004


def project(points, cams):
    """Vectorised BAL projection of points (n,3) by cameras (n,9) in the reference's camera layout
    (r, t, k1, k2, f).  Data generation only; parity checks use oracle/."""
    r, t = cams[:, :3], cams[:, 3:6]
    k1, k2, f = cams[:, 6], cams[:, 7], cams[:, 8]
    th = np.sqrt((r * r).sum(1))[:, None]
    k = r / th
    c, s = np.cos(th), np.sin(th)
    d = (k * points).sum(1)[:, None]
    P1 = c * points + s * np.cross(k, points) + (1 - c) * d * k + t
    P2 = -P1[:, :2] / P1[:, 2:3]
    n = (P2 * P2).sum(1)
    return (f * (1.0 + k1 * n + k2 * n * n))[:, None] * P2


def _degrees(rng, npnts, nobs, ncams):
    lo, hi = 2, ncams
    if nobs < lo * npnts or nobs > hi * npnts:
        raise ValueError("nobs must lie in [2*npnts, ncams*npnts]")
    mean_extra = nobs / npnts - lo
    if mean_extra <= 0:
        deg = np.full(npnts, lo, dtype=np.int64)
    else:
        pgeo = 1.0 / (1.0 + mean_extra)
        deg = lo + rng.geometric(pgeo, size=npnts).astype(np.int64) - 1
        deg = np.minimum(deg, hi)
    diff = int(nobs - deg.sum())
    while diff != 0:
        step = 1 if diff > 0 else -1
        cand = np.flatnonzero(deg < hi) if step > 0 else np.flatnonzero(deg > lo)
        take = min(abs(diff), len(cand))
        pick = rng.choice(cand, size=take, replace=False)
        deg[pick] += step
        diff -= step * take
    return deg


def _cameras_in_the_plane(rng, ncams, npnts, nobs, radius):
    """Observation graph of a scene laid out in the plane (street-level captures of a town): every camera has a random position
    in the unit square -- unrelated to its NUMBER -- and a point is seen by cameras among the K = ncams * pi * radius^2 nearest
    to a random centre.  Two cameras share points when they stand within ~2 radius of each other: the camera graph is a
    two-dimensional geometric graph whose numbering carries no structure.  -> (pnt0, cam0) sorted by point then camera."""
    from scipy.spatial import cKDTree
    pos = rng.random((ncams, 2))
    K = int(min(ncams, max(2, round(ncams * np.pi * radius * radius))))
    deg = _degrees(rng, npnts, nobs, K)
    tree = cKDTree(pos)
    cam0 = np.empty(nobs, dtype=np.int64)
    start = np.concatenate([[0], np.cumsum(deg)])
    for lo in range(0, npnts, 200000):  # chunks: the candidate table is (points, K)
        hi = min(npnts, lo + 200000)
        _, near = tree.query(rng.random((hi - lo, 2)), k=K)
        near = near.reshape(hi - lo, K)
        pick = np.argsort(rng.random((hi - lo, K)), axis=1)  # a random subset of the candidates: the first deg of a shuffle
        d = deg[lo:hi]
        mask = np.arange(K)[None, :] < d[:, None]
        cam0[start[lo]:start[hi]] = np.take_along_axis(near, pick, axis=1)[mask]
    pnt0 = np.repeat(np.arange(npnts, dtype=np.int64), deg)
    order = np.lexsort((cam0, pnt0))
    return pnt0, cam0[order]


def make_problem(ncams, npnts, nobs, seed=BASE_SEED, locality=None, plane_radius=None):
    """-> dict(cam_idx1, pnt_idx1, pt2d, x0, x_true, ncams, npnts, nobs) in the reference's conventions.

    plane_radius (None or a radius in the unit square): the cameras stand in the plane and a point is seen from one
    neighbourhood (_cameras_in_the_plane): a sparse reduced camera matrix WITHOUT a band in the numbering it comes with.

    locality (None or a fraction in (0, 1]): None draws the cameras of a point uniformly from ALL cameras -- every camera
    pair then shares points and the reduced camera matrix S is dense (its block fill is 1).  With locality = w a point is
    seen only by cameras of one window of round(w * ncams) consecutive cameras (window start uniform): cameras further
    apart than the window share no point, S is block-banded with half bandwidth w * ncams and its 9 x 9 block fill is
    about 2w - w^2 (schur_fill() measures it).  Real BAL problems lie in between (sequential captures: narrow band plus loop
    closures); the default stays dense -- the worst case for the factorisation."""
    rng = np.random.default_rng(seed)
    # points in the unit ball
    g = rng.standard_normal((npnts, 3))
    g /= np.linalg.norm(g, axis=1)[:, None]
    pts = g * rng.random((npnts, 1)) ** (1.0 / 3.0)
    # cameras
    axis = rng.standard_normal((ncams, 3))
    axis /= np.linalg.norm(axis, axis=1)[:, None]
    ang = rng.uniform(0.05, 1.5, size=(ncams, 1))
    rvec = axis * ang
    rho = rng.uniform(5.0, 7.0, size=ncams)
    tvec = np.zeros((ncams, 3))
    tvec[:, 2] = -rho  # P1 = R X - rho e_z  => P1.z in (-8, -4)
    f = rng.uniform(400.0, 1800.0, size=ncams)
    k1 = -rng.uniform(0.0, 5.0, size=ncams) * 1e-7
    k2 = rng.uniform(0.0, 1.0, size=ncams) * 1e-12
    cams = np.column_stack([rvec, tvec, k1, k2, f])
    # observation graph
    if plane_radius is not None:
        if locality is not None:
            raise ValueError("locality and plane_radius are two different observation graphs")
        pnt0, cam0 = _cameras_in_the_plane(rng, ncams, npnts, nobs, float(plane_radius))
        lo_cam, width = None, 0
    elif locality is None:
        deg = _degrees(rng, npnts, nobs, ncams)
        pnt0 = np.repeat(np.arange(npnts, dtype=np.int64), deg)
        lo_cam = np.zeros(nobs, dtype=np.int64)
        width = ncams
    else:
        width = int(min(ncams, max(2, round(float(locality) * ncams))))
        deg = _degrees(rng, npnts, nobs, width)
        pnt0 = np.repeat(np.arange(npnts, dtype=np.int64), deg)
        lo_cam = np.repeat(rng.integers(0, ncams - width + 1, size=npnts, dtype=np.int64), deg)  # window start of the point
    if lo_cam is not None:
        cam0 = lo_cam + rng.integers(0, width, size=nobs, dtype=np.int64)
    for _ in range(400 if lo_cam is not None else 0):
        order = np.lexsort((cam0, pnt0))
        cam0, lo_cam = cam0[order], lo_cam[order]
        dup = np.flatnonzero((pnt0[1:] == pnt0[:-1]) & (cam0[1:] == cam0[:-1])) + 1
        if dup.size == 0:
            break
        cam0[dup] = lo_cam[dup] + rng.integers(0, width, size=dup.size, dtype=np.int64)
    else:
        if lo_cam is not None:
            raise RuntimeError("could not draw distinct cameras per point")
    proj = project(pts[pnt0], cams[cam0])
    pt2d = (proj + rng.normal(0.0, 0.5, size=proj.shape)).ravel()
    x_true = np.concatenate([pts.ravel(), cams.ravel()])
    pts0 = pts + rng.normal(0.0, 0.02, size=pts.shape)
    cams0 = cams * (1.0 + rng.normal(0.0, 1e-3, size=cams.shape))
    x0 = np.concatenate([pts0.ravel(), cams0.ravel()])
    return dict(cam_idx1=cam0 + 1, pnt_idx1=pnt0 + 1, pt2d=np.ascontiguousarray(pt2d), x0=x0, x_true=x_true,
                ncams=int(ncams), npnts=int(npnts), nobs=int(nobs))


def shuffle_cameras(prob, seed=0, sigma=None):
    """The same problem with its cameras renumbered at random: (problem, sigma) with camera c of `prob` called sigma[c]
    (0-based) in the result; observations stay in BAL order (by point, then by the NEW camera number).  A BAL file promises
    nothing about how its cameras are numbered: this is what a fill-reducing ordering (`perm`, src/lm.jl:84-88) has to undo.
    Camera blocks of x map as x_new[3 npnts + 9 sigma[c] + j] = x[3 npnts + 9 c + j]; points keep their numbers.
    sigma: an explicit renumbering instead of a random one."""
    ncams, npnts, nobs = prob["ncams"], prob["npnts"], prob["nobs"]
    sigma = (np.random.default_rng(seed).permutation(ncams) if sigma is None else np.asarray(sigma)).astype(np.int64)
    assert sorted(sigma.tolist()) == list(range(ncams))
    cam0 = sigma[np.asarray(prob["cam_idx1"]) - 1]
    pnt0 = np.asarray(prob["pnt_idx1"]) - 1
    order = np.lexsort((cam0, pnt0))
    out = dict(prob)
    out["cam_idx1"] = cam0[order] + 1
    out["pnt_idx1"] = pnt0[order] + 1
    out["pt2d"] = np.ascontiguousarray(np.asarray(prob["pt2d"]).reshape(nobs, 2)[order].ravel())
    for key in ("x0", "x_true"):
        x = np.array(prob[key], copy=True)
        cams = x[3 * npnts:].reshape(ncams, 9)
        new = np.empty_like(cams)
        new[sigma] = cams
        x[3 * npnts:] = new.ravel()
        out[key] = x
    return out, sigma


def unshuffle_vector(v, sigma, npnts):
    """A vector in the layout of x of a shuffle_cameras problem, back in the original camera numbering."""
    v = np.array(v, copy=True)
    cams = v[3 * npnts:].reshape(len(sigma), 9)
    v[3 * npnts:] = cams[sigma].ravel()
    return v


def schur_fill(prob, tile=128):
    """(block fill, tile fill) of the reduced camera matrix of a problem: the fraction of camera pairs (a >= b) that share at
    least one point, and the fraction of the lower tile x tile tiles of S (rows / columns 9 * camera) that hold such a pair
    -- before factorisation (the factor fills in inside the profile)."""
    ncams = prob["ncams"]
    cam0 = np.asarray(prob["cam_idx1"]) - 1
    pnt0 = np.asarray(prob["pnt_idx1"]) - 1
    order = np.lexsort((cam0, pnt0))
    cam0, pnt0 = cam0[order], pnt0[order]
    start = np.flatnonzero(np.r_[True, pnt0[1:] != pnt0[:-1]])
    deg = np.diff(np.r_[start, len(pnt0)])
    pairs = set()
    share = np.zeros((ncams, ncams), dtype=bool) if ncams <= 20000 else None
    if share is None:
        raise ValueError("schur_fill: too many cameras for the dense indicator")
    for d in np.unique(deg):  # all points of one degree at once
        idx = start[deg == d]
        cams = cam0[idx[:, None] + np.arange(d)[None, :]]  # (npoints_d, d), ascending per row
        for a in range(d):
            for b in range(a + 1):
                share[cams[:, a], cams[:, b]] = True
    nblk = ncams * (ncams + 1) // 2
    block_fill = float(np.count_nonzero(np.tril(share))) / nblk
    nt = (9 * ncams + tile - 1) // tile
    occ = np.zeros((nt, nt), dtype=bool)
    ia, ib = np.nonzero(np.tril(share))
    for da in (0, 8):
        for db in (0, 8):
            occ[(9 * ia + da) // tile, (9 * ib + db) // tile] = True
    tile_fill = float(np.count_nonzero(np.tril(occ))) / (nt * (nt + 1) // 2)
    return block_fill, tile_fill


def make_named(name, seed_offset=0, scale=1.0, locality=None, plane_radius=None):
    """One of SHAPES, optionally shrunk by `scale` (observations per point kept)."""
    ncams, npnts, nobs = SHAPES[name]
    if scale != 1.0:
        ncams = max(4, int(round(ncams * scale)))
        npnts2 = max(8, int(round(npnts * scale)))
        nobs = max(2 * npnts2, int(round(nobs * npnts2 / npnts)))
        nobs = min(nobs, ncams * npnts2)
        npnts = npnts2
    return make_problem(ncams, npnts, nobs, BASE_SEED + seed_offset, locality=locality, plane_radius=plane_radius)


def as_arrays(prob, T=np.float64):
    """tuple in the order readfile() returns (src/ReadFiles.jl:52)."""
    return (prob["cam_idx1"], prob["pnt_idx1"], prob["pt2d"].astype(T), prob["x0"].astype(T), prob["ncams"],
            prob["npnts"], prob["nobs"])


def write_bal(path, prob, x=None):
    """Write a problem in BAL text format (bzip2 when the name ends in .bz2): 0-based indices, camera order
    r t f k1 k2 (the reader restores r t k1 k2 f, src/ReadFiles.jl:32-43).  %.17g round-trips Float64."""
    x = prob["x0"] if x is None else x
    ncams, npnts, nobs = prob["ncams"], prob["npnts"], prob["nobs"]
    pts = x[: 3 * npnts]
    cams = x[3 * npnts:].reshape(ncams, 9)
    lines = [f"{ncams} {npnts} {nobs}"]
    p2 = prob["pt2d"].reshape(nobs, 2)
    for k in range(nobs):
        lines.append(f"{prob['cam_idx1'][k] - 1} {prob['pnt_idx1'][k] - 1}     {p2[k, 0]:.17g} {p2[k, 1]:.17g}")
    for c in cams:
        for v in (c[0], c[1], c[2], c[3], c[4], c[5], c[8], c[6], c[7]):
            lines.append(f"{v:.17g}")
    for v in pts:
        lines.append(f"{v:.17g}")
    data = ("\n".join(lines) + "\n").encode()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.endswith(".bz2"):
        with bz2.open(path, "wb") as fh:
            fh.write(data)
    else:
        with open(path, "wb") as fh:
            fh.write(data)
    return path
