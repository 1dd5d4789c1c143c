"""Fill-reducing camera ordering of the reduced camera system (`perm` = :AMD / :Metis of the reference's solvers,
src/lm.jl:84-88, src/LevenbergMarquardt.jl:106-110, consumed by ldl_analyse, src/ldl_aux.jl:246-283) -- host logic, no GPU.

AMD.jl / Metis.jl are third-party C libraries absent from the image: what is tested is the ROLE of an ordering (a
permutation, deterministic, that keeps the tile pattern of a badly numbered problem close to that of a well numbered one);
its arithmetic effect on the step is covered by the GPU parity tests.  Parity of the sequences themselves: unpinned."""
import numpy as np
import pytest


def _band_problem(ba, ncams=420, npnts=5000, nobs=22000, locality=0.1, seed=31):
    return ba.synthetic.make_problem(ncams, npnts, nobs, seed=seed, locality=locality)


@pytest.mark.parametrize("method", ["AMD", "Metis"])
def test_ordering_undoes_a_random_camera_numbering(ba, method):
    p = _band_problem(ba)
    q, sigma = ba.synthetic.shuffle_cameras(p, seed=5)
    _, tf_nat, ff_nat, bf = ba.schur_ordering(p["cam_idx1"], p["pnt_idx1"], p["ncams"], p["npnts"], "natural")
    _, tf_shuf, ff_shuf, bf2 = ba.schur_ordering(q["cam_idx1"], q["pnt_idx1"], q["ncams"], q["npnts"], "natural")
    perm, tf, ff, bf3 = ba.schur_ordering(q["cam_idx1"], q["pnt_idx1"], q["ncams"], q["npnts"], method)
    assert abs(bf - bf2) < 1e-15 and abs(bf - bf3) < 1e-15, "the block fill does not depend on the numbering"
    assert sorted(perm.tolist()) == list(range(1, p["ncams"] + 1)), "not a permutation"
    print(f"{method}: block fill {bf:.3f}; tile fill as generated {tf_nat:.3f}, shuffled {tf_shuf:.3f}, shuffled + ordering {tf:.3f}; "
          f"update tiles / dense {ff_nat:.4f}, {ff_shuf:.4f}, {ff:.4f}")
    assert tf_shuf > 0.9, "a random numbering should fill the tile pattern"
    assert tf <= 1.25 * tf_nat, f"tile fill {tf:.3f} with the ordering against {tf_nat:.3f} for the generator's numbering"
    assert ff <= 1.5 * ff_nat + 0.01
    # deterministic
    perm2, tf2, ff2, _ = ba.schur_ordering(q["cam_idx1"], q["pnt_idx1"], q["ncams"], q["npnts"], method)
    assert np.array_equal(perm, perm2) and tf == tf2 and ff == ff2


def test_a_well_numbered_problem_keeps_its_numbering_or_improves(ba):
    p = _band_problem(ba)
    _, tf_nat, ff_nat, _ = ba.schur_ordering(p["cam_idx1"], p["pnt_idx1"], p["ncams"], p["npnts"], "natural")
    for method in ("AMD", "Metis"):
        perm, tf, ff, _ = ba.schur_ordering(p["cam_idx1"], p["pnt_idx1"], p["ncams"], p["npnts"], method)
        # the caller's numbering is a candidate: an ordering is never costlier than it.  (Cost, not fill: a sequence that
        # eliminates a long profile from both ends -- two chains side by side -- may pay a few tiles for half the chain.)
        assert ff <= 1.25 * ff_nat + 1e-15 and tf <= 1.1 * tf_nat
    perm, _, _, _ = ba.schur_ordering(p["cam_idx1"], p["pnt_idx1"], p["ncams"], p["npnts"], "natural")
    assert np.array_equal(perm, np.arange(1, p["ncams"] + 1))


def test_hub_cameras_are_deferred(ba):
    """A few cameras that see points all over the scene (an overview shot) connect every window of a sequential capture:
    breadth-first sequences collapse into two levels unless the hubs go last."""
    rng = np.random.default_rng(3)
    p = _band_problem(ba, locality=0.06)
    ncams, npnts = p["ncams"], p["npnts"]
    hubs = rng.choice(ncams, size=6, replace=False)
    extra_p = rng.choice(npnts, size=3000, replace=False)
    extra_c = hubs[rng.integers(0, len(hubs), size=len(extra_p))]
    cam = np.concatenate([p["cam_idx1"], extra_c + 1])
    pnt = np.concatenate([p["pnt_idx1"], extra_p + 1])
    key = np.unique(pnt.astype(np.int64) * (ncams + 1) + cam)  # (point, camera) pairs, duplicates dropped, BAL order
    cam, pnt = key % (ncams + 1), key // (ncams + 1)
    sigma = rng.permutation(ncams)
    cam_s = sigma[cam - 1] + 1
    _, tf_band, _, _ = ba.schur_ordering(p["cam_idx1"], p["pnt_idx1"], ncams, npnts, "natural")
    for method in ("AMD", "Metis"):
        perm, tf, ff, _ = ba.schur_ordering(cam_s, pnt, ncams, npnts, method)
        print(f"{method}: tile fill {tf:.3f} with 6 hub cameras (the band alone: {tf_band:.3f})")
        assert tf <= tf_band + 0.2, "hub cameras must not fill the whole pattern"
        tail = set((perm[-12:] - 1).tolist())
        assert len(tail & set(sigma[hubs].tolist())) >= 5, "the hub cameras belong at the end of the sequence"


def test_ordering_edge_cases(ba):
    # two cameras, one point; a camera without observations; disconnected components
    perm, tf, ff, bf = ba.schur_ordering(np.array([1, 2]), np.array([1, 1]), 2, 1, "AMD")
    assert sorted(perm.tolist()) == [1, 2]
    cam = np.array([1, 2, 4, 5, 1, 2], dtype=np.int64)
    pnt = np.array([1, 1, 2, 2, 3, 3], dtype=np.int64)
    for method in ("AMD", "Metis", "natural"):
        perm, tf, ff, bf = ba.schur_ordering(cam, pnt, 6, 3, method)
        assert sorted(perm.tolist()) == [1, 2, 3, 4, 5, 6]
    with pytest.raises(ba.BAError):
        ba.schur_ordering(np.array([3]), np.array([1]), 2, 1, "AMD")  # camera index out of range


def test_two_ended_elimination_of_a_long_profile(ba):
    """From 768 cameras on, the best profile sequence is also offered eliminated FROM BOTH ENDS (first half forwards, the
    cameras not adjacent to it backwards from the far end, the frontier last): two independent runs of tile column pairs, no
    fill beyond the band's own.  The fill must stay that of the one-ended sequence; that the runs are independent is checked
    by the symbolic phase itself (and the factorisation's results by the GPU tests)."""
    p = ba.synthetic.make_problem(1100, 6000, 30000, seed=35, locality=0.1)
    q, _ = ba.synthetic.shuffle_cameras(p, seed=8)
    _, tf_nat, ff_nat, _ = ba.schur_ordering(p["cam_idx1"], p["pnt_idx1"], p["ncams"], p["npnts"], "natural")
    for prob in (p, q):
        perm, tf, ff, _ = ba.schur_ordering(prob["cam_idx1"], prob["pnt_idx1"], prob["ncams"], prob["npnts"], "AMD")
        assert sorted(perm.tolist()) == list(range(1, 1101))
        assert tf <= 1.25 * tf_nat, (tf, tf_nat)


def test_cameras_in_the_plane(ba):
    """A scene laid out in the plane (synthetic.make_problem(plane_radius=...)): the camera graph is a two-dimensional
    geometric graph and the cameras' numbers carry no structure at all -- as numbered, every tile of S is occupied.  Either
    method must bring the tile pattern down to a fraction (the sequences compete at tile granularity: DESIGN 5c)."""
    p = ba.synthetic.make_problem(900, 9000, 45000, seed=41, plane_radius=0.09)
    cam, pnt = p["cam_idx1"], p["pnt_idx1"]
    same = pnt[1:] == pnt[:-1]
    assert np.all(np.diff(pnt) >= 0) and np.all(cam[1:][same] > cam[:-1][same])  # BAL order, a camera once per point
    assert np.bincount(pnt - 1, minlength=p["npnts"]).min() >= 2
    _, tf_nat, ff_nat, bf = ba.schur_ordering(cam, pnt, p["ncams"], p["npnts"], "natural")
    assert tf_nat > 0.95 and bf < 0.15
    for method in ("AMD", "Metis"):
        perm, tf, ff, _ = ba.schur_ordering(cam, pnt, p["ncams"], p["npnts"], method)
        print(f"{method}: tile fill {tf:.3f} (as numbered {tf_nat:.3f}), update flops {ff:.3f} of the dense factorisation's; block fill {bf:.3f}")
        assert sorted(perm.tolist()) == list(range(1, p["ncams"] + 1))
        assert tf < 0.5 and ff < 0.2
        perm2, _, _, _ = ba.schur_ordering(cam, pnt, p["ncams"], p["npnts"], method)
        assert np.array_equal(perm, perm2)
