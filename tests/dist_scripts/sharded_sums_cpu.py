"""Run by torch.distributed.run with 2 ranks (gloo, CPU only): the sharding logic (partition_by_point, shard_problem)
and the cross-rank sums of the design (camera-side J'r, Hcc, reduced camera matrix S, rhs) emulated with the oracle's
J and r in numpy; after the all-reduce every rank must hold the unsharded quantities and the Schur step must equal
the oracle's augmented-system step."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch.distributed as dist


def blocks(orc, arrays, lam, add_diag):
    cam, pnt, pt2d, x, ncams, npnts, nobs = arrays
    r = orc.residuals(cam, pnt, x, pt2d, npnts).reshape(-1, 2)
    J = orc.jac_coord(cam, pnt, x, npnts).reshape(-1, 2, 12)
    A, B = J[:, :, :3], J[:, :, 3:]
    n = 9 * ncams
    Hpp = np.zeros((npnts, 3, 3)); gp = np.zeros((npnts, 3)); Hcc = np.zeros((ncams, 9, 9)); gc = np.zeros((ncams, 9))
    np.add.at(Hpp, pnt - 1, np.einsum("kai,kaj->kij", A, A)); np.add.at(gp, pnt - 1, np.einsum("kai,ka->ki", A, r))
    np.add.at(Hcc, cam - 1, np.einsum("kai,kaj->kij", B, B)); np.add.at(gc, cam - 1, np.einsum("kai,ka->ki", B, r))
    Uinv = np.linalg.inv(Hpp + lam * np.eye(3)[None])
    S = np.zeros((n, n)); rhs = -gc.reshape(-1).copy()
    for c in range(ncams):
        S[9 * c:9 * c + 9, 9 * c:9 * c + 9] += Hcc[c] + (lam * np.eye(9) if add_diag else 0)
    W = np.einsum("kai,kaj->kij", A, B)  # 3x9 per obs
    order = np.argsort(pnt, kind="stable")
    start = 0
    while start < nobs:
        p = pnt[order[start]]; end = start
        while end < nobs and pnt[order[end]] == p: end += 1
        ks = order[start:end]
        U = Uinv[p - 1]
        for a in ks:
            rhs[9 * (cam[a] - 1):9 * cam[a]] += W[a].T @ (U @ gp[p - 1])
            for b in ks:
                S[9 * (cam[a] - 1):9 * cam[a], 9 * (cam[b] - 1):9 * cam[b]] -= W[a].T @ U @ W[b]
        start = end
    return dict(S=S, rhs=rhs, gc=gc.reshape(-1), gp=gp, Uinv=Uinv, W=W, pnt=pnt, cam=cam)


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ba = ge.load_package(); orc = ge.load_oracle()
    prob = ba.synthetic.make_problem(7, 80, 330, seed=9)
    arrays = ba.synthetic.as_arrays(prob)
    lam = 3.0
    local, info = ba.parallel.shard_problem(arrays, rank, world)
    loc = blocks(orc, local, lam, add_diag=(rank == 0))   # lambda I of the camera block is added once (rank 0)
    S = ba.parallel.allreduce_sum_numpy(loc["S"]); rhs = ba.parallel.allreduce_sum_numpy(loc["rhs"])
    gc = ba.parallel.allreduce_sum_numpy(loc["gc"])
    dc = np.linalg.solve(S, rhs)
    # local back-substitution, then assemble the global step
    npl = local[5]
    dp = np.zeros((npl, 3))
    acc = np.zeros((npl, 3))
    for k in range(local[6]):
        acc[loc["pnt"][k] - 1] += loc["W"][k] @ dc[9 * (loc["cam"][k] - 1):9 * loc["cam"][k]]
    for p in range(npl):
        dp[p] = -loc["Uinv"][p] @ (loc["gp"][p] + acc[p])
    delta = ba.parallel.gather_solution(np.concatenate([dp.ravel(), dc]), info, prob["ncams"])
    out = None
    if rank == 0:
        full = blocks(orc, arrays, lam, add_diag=True)
        rc, d_ref, _, jtr_ref = orc.lm_step(prob["ncams"], prob["npnts"], prob["cam_idx1"], prob["pnt_idx1"], prob["pt2d"], prob["x0"], lam)
        out = dict(S=float(np.abs(S - full["S"]).max() / np.abs(full["S"]).max()),
                   rhs=float(np.abs(rhs - full["rhs"]).max() / np.abs(full["rhs"]).max()),
                   gc=float(np.abs(gc - jtr_ref[3 * prob["npnts"]:]).max() / np.abs(jtr_ref).max()),
                   delta=float(np.linalg.norm(delta - d_ref) / np.linalg.norm(d_ref)), rc=rc)
        json.dump(out, open(sys.argv[1], "w"))
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
