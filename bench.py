#!/usr/bin/env python3
"""bench.py -- LM iterations/s (+ Jacobian Mnnz/s) of the MI355X hot path on a synthetic BAL-shaped problem.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one Levenberg-Marquardt iteration of src/lm.jl (factor + solve of the damped normal equations, trial
residual, accept/reject; accepted steps add a Jacobian refresh), run with all stopping tolerances at zero so that
exactly K iterations execute.  Workload at every N: Venice-1778-993923-shaped synthetic data (BASELINE.json's metric),
observations sharded by point over the N ranks, cameras replicated => "strong" scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import os
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # oracle baseline: idle OpenMP threads must not spin
import ctypes as C
import json
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # FP32 matrix peak (v_mfma_f32_16x16x4_f32: twice the f64 rate)
MFMA_F64_PEAK_TF = 78.6    # MI355X FP64 matrix peak (vendor sheet; v_mfma_f64_16x16x4_f64, 2048 flop / 64 clk / SIMD)


def kernel_sources_sha():
    """sha256 over the device sources (csrc/*.hip, csrc/*.h, csrc/*.cpp, in name order): what a PMC measurement is a
    measurement OF.  tools/pmc_traffic.py stamps the file it writes with it."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "bundleadjustment.jl_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this workload (profiles/pmc_traffic_<workload>.json,
    written by tools/collect_pmc.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this very command; the
    counters cannot be read from inside the timed process).  None when no such measurement is committed -- or when it was
    taken on other kernel sources than the ones this run was built from (the file carries their sha256): a stale number is
    not reported."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    try:
        with open(path) as f:
            d = json.load(f)
        k = d["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    if d.get("kernel_sources_sha256") != kernel_sources_sha():
        return None
    for name, v in k.items():
        if name.startswith(kernel):
            return v["hbm_bytes_per_launch"]
    return None


def pmc_stamp(workload):
    """what the bench line says about where `traffic` comes from"""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return {"file": None}
    now = kernel_sources_sha()
    return {"file": os.path.relpath(path, ROOT), "taken_at_commit": d.get("commit"), "kernel_sources_sha256": (d.get("kernel_sources_sha256") or "")[:16],
            "current_sources_sha256": now[:16], "current": d.get("kernel_sources_sha256") == now}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="venice-1778")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debugging only; invalid as a result)")
    ap.add_argument("--locality", type=float, default=None,
                    help="cameras of a point drawn from a window of this fraction of the cameras: block-banded reduced camera "
                         "matrix (synthetic.make_problem); default: every camera pair shares points, dense S")
    ap.add_argument("--plane-radius", type=float, default=None,
                    help="cameras standing in the plane, a point seen from one neighbourhood of this radius (unit square): a sparse "
                         "reduced camera matrix with NO band in the numbering it comes with (synthetic._cameras_in_the_plane)")
    ap.add_argument("--shuffle-cameras", action="store_true",
                    help="renumber the cameras at random (synthetic.shuffle_cameras): what `perm` has to undo (not the headline configuration)")
    ap.add_argument("--perm", choices=["AMD", "Metis", "natural"], default="AMD",
                    help="fill-reducing camera ordering of the reduced camera system (src/lm.jl:84-88); natural = the caller's numbering")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--cpu-full", default="dubrovnik-356",
                    help="workload of the fully timed CPU iteration reported as cpu_baseline_full beside the GPU's time on the "
                         "same shape (N = 1 only; 'none' skips it)")
    ap.add_argument("--facto-type", choices=["f64", "f32"], default="f64",
                    help="facto_type of lm.jl (f32 = the diffprecsions.jl path, BASELINE config 5); the default line is f64")
    ap.add_argument("--facto", choices=["ldl", "pcg"], default="ldl",
                    help="ldl: the reference's direct branch (the default line); pcg: the matrix-free CG extension "
                         "(--pcg-tol: its relative residual)")
    ap.add_argument("--pcg-tol", type=float, default=1e-8)
    ap.add_argument("--no-pcg", action="store_true", help="skip the secondary facto = :PCG measurement")
    ap.add_argument("--emulate-shard", default=None, metavar="R/W",
                    help="timing rehearsal on one GPU: run shard R of W of the workload (that rank's observations and points, all "
                         "cameras) WITHOUT a communicator -- what one rank of a W-GPU job computes per iteration; the numbers "
                         "are a rank's compute time, the results are not a solution of the full problem")
    ap.add_argument("--backend", default=os.environ.get("BA_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend: nccl (= RCCL, default) or gloo (host-staged all-reduce; rehearsal only)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (several ranks sharing one GPU need --backend gloo)")
    return ap.parse_args()


FACTO_TYPE = None  # set from --facto-type
FACTO, PCG_TOL = "LDL", None  # set from --facto / --pcg-tol
PERM = "AMD"                   # set from --perm


def lm_fixed_iterations(ba, fr, k, x=None, x_device_ptr=None):
    """exactly k iterations of lm.jl: every stopping test disabled except the iteration cap.  x_device_ptr: the iterate lives
    in device memory (ba_lm_solve_dev): no host copy of x inside the call"""
    return ba.Levenberg_Marquardt(fr, FACTO, PERM, "None", False, x=x, ite_max=k - 1, restol=0.0, satol=0.0, srtol=0.0,
                                  oatol=0.0, ortol=0.0, atol=0.0, rtol=0.0, log=False, facto_type=FACTO_TYPE, pcg_tol=PCG_TOL,
                                  x_device_ptr=x_device_ptr)


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N rank processes (torch.distributed.run, one per
    GPU, rendezvous on 127.0.0.1) from THIS process, which has not touched the GPU, and leave with their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    global FACTO_TYPE, FACTO, PCG_TOL, PERM
    args = parse()
    PERM = args.perm
    if args.facto == "pcg":
        FACTO, PCG_TOL = "PCG", args.pcg_tol
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.facto_type == "f32":
        import numpy as _np
        FACTO_TYPE = _np.float32
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] launched with WORLD_SIZE={world} but --gpus {args.gpus}: running {world} rank(s)", file=sys.stderr)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")  # initialise torch's HIP context before c10d counts the devices
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=args.backend)
    ba = ge.load_package()

    # ---- workload ---------------------------------------------------------------------------------------------------
    t0 = time.time()
    prob = ba.synthetic.make_named(args.workload, scale=args.scale, locality=args.locality, plane_radius=args.plane_radius)
    if args.shuffle_cameras:
        prob, _ = ba.synthetic.shuffle_cameras(prob, seed=ba.synthetic.BASE_SEED)
    arrays = ba.synthetic.as_arrays(prob)
    if world > 1:
        arrays, info = ba.parallel.shard_problem(arrays, rank, world)
    elif args.emulate_shard:
        er, ew = (int(v) for v in args.emulate_shard.split("/"))
        arrays, info = ba.parallel.shard_problem(arrays, er, ew)
    nlp = ba.BALNLPModel(arrays=arrays, device=local_rank, model_name=args.workload)
    fr = ba.FeasibilityResidual(nlp)
    reducer = ba.parallel.CameraBlockReducer(nlp) if world > 1 else None
    t_setup = time.time() - t0
    nobs_g, npnts_g, ncams = prob["nobs"], prob["npnts"], prob["ncams"]
    nvar_g = 3 * npnts_g + 9 * ncams

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (also allocates the LM workspace and builds the Schur task list) ---------------------------------------
    # The timed region starts with its input resident in HBM: x0 sits in device memory and the K iterations run through
    # ba_lm_solve_dev.  (The reference's boundary hands over host vectors; the rate through ba_lm_solve, with the upload of x0
    # and the download of the solution inside the call, is measured once below and reported as `host_boundary`.)
    x0_dev = torch.from_numpy(np.ascontiguousarray(arrays[3], dtype=np.float64)).cuda()
    x_work = torch.empty_like(x0_dev)
    if args.warmup > 0:
        x_work.copy_(x0_dev)
        lm_fixed_iterations(ba, fr, args.warmup, x_device_ptr=x_work.data_ptr())
    x_work.copy_(x0_dev)
    barrier()
    t1 = time.perf_counter()
    st = lm_fixed_iterations(ba, fr, args.steps, x_device_ptr=x_work.data_ptr())
    barrier()
    elapsed = time.perf_counter() - t1
    t1 = time.perf_counter()
    st_host = lm_fixed_iterations(ba, fr, args.steps)  # the same K iterations from / to host memory
    barrier()
    elapsed_host = time.perf_counter() - t1
    assert st_host.iter == st.iter and st_host.objective == st.objective, (st_host.objective, st.objective)
    def max_over_ranks(v):
        if world <= 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = max_over_ranks(elapsed)
    elapsed_host = max_over_ranks(elapsed_host)
    assert st.iter == args.steps, (st.iter, args.steps)

    # ---- the same K iterations by the matrix-free CG extension (facto = :PCG, relative residual 1e-8), reported beside the
    # headline (which stays on the reference's direct branch): the path that scales with the number of GPUs ------------------
    pcg_line = None
    if FACTO != "PCG" and not args.no_pcg:
        def lm_pcg(k):
            return ba.Levenberg_Marquardt(fr, "PCG", "AMD", "None", False, ite_max=k - 1, restol=0.0, satol=0.0, srtol=0.0, oatol=0.0,
                                          ortol=0.0, atol=0.0, rtol=0.0, log=False, pcg_tol=1e-8)
        lm_pcg(max(1, min(2, args.warmup)))
        barrier()
        t2 = time.perf_counter()
        stp = lm_pcg(args.steps)
        barrier()
        el2 = max_over_ranks(time.perf_counter() - t2)
        pcg_line = {"metric": "LM iterations/s, facto = :PCG (pcg_tol 1e-8)", "value": stp.iter / el2, "ms_per_step": 1e3 * el2 / max(1, stp.iter),
                    "n_cg": stp.n_cg, "objective": stp.objective, "objective_direct": st.objective}

    # ---- Jacobian throughput: jac_coord on device-resident x / vals, events on the launch stream ---------------------
    stream = torch.cuda.Stream()  # a real (non-null) stream: the library treats a null stream as "the handle's own"
    x_dev = torch.from_numpy(np.ascontiguousarray(arrays[3])).cuda()
    vals_dev = torch.empty(24 * nlp.nobs, dtype=torch.float64, device="cuda")
    L = ba._lib.lib()
    sp = C.c_void_p(stream.cuda_stream)

    def jac():
        ba._lib.check(L.ba_jac_coord_dev(nlp.handle, C.c_void_p(x_dev.data_ptr()), C.c_void_p(vals_dev.data_ptr()), sp))

    torch.cuda.synchronize()
    for _ in range(10):
        jac()
    # 60 back-to-back launches (round 2: 20, which on a box that has just idled through the host-side set-up above mostly
    # measured the clock ramp: 61 us per residual launch against 47-49 at steady state, tools/bench_res.py)
    reps = 60
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    e0.record(stream)
    for _ in range(reps):
        jac()
    e1.record(stream)
    torch.cuda.synchronize()
    jac_ms = e0.elapsed_time(e1) / reps
    jac_ms = max_over_ranks(jac_ms)
    jac_mnnz = 24.0 * nobs_g / (jac_ms * 1e-3) / 1e6
    # algorithmic bytes of jac_coord!: 2 int64 indices + 24 doubles out + every parameter read once
    jac_bytes = (208.0 + 8.0 * nvar_g / nobs_g) * nobs_g / world
    jac_gbs = jac_bytes / (jac_ms * 1e-3) / 1e9

    # ---- residual throughput (cons!): the same way; one launch = k_cam_pre (a few us) + k_residual --------------------------
    r_dev = torch.empty(2 * nlp.nobs, dtype=torch.float64, device="cuda")

    def res():
        ba._lib.check(L.ba_residual_dev(nlp.handle, C.c_void_p(x_dev.data_ptr()), C.c_void_p(r_dev.data_ptr()), sp))

    for _ in range(10):
        res()
    barrier()
    e0.record(stream)
    for _ in range(reps):
        res()
    e1.record(stream)
    torch.cuda.synchronize()
    res_ms = max_over_ranks(e0.elapsed_time(e1) / reps)
    # algorithmic bytes of cons!: 2 int64 indices + 2 doubles of pt2d + 2 doubles out + every parameter once
    res_bytes = (48.0 + 8.0 * nvar_g / nobs_g) * nobs_g / world
    res_gbs = res_bytes / (res_ms * 1e-3) / 1e9
    del r_dev

    # ---- per-kernel profile of the same K iterations (separate, synchronising run) ---------------------------------------
    prof = {}
    if not args.no_profile:
        nlp.profile(True)
        lm_fixed_iterations(ba, fr, args.steps)
        prof = nlp.profile_get()
        nlp.profile(False)
    roof = None
    if prof:
        name, (ms, calls) = max(((k, v) for k, v in prof.items() if k != "allreduce"), key=lambda kv: kv[1][0])
        n_fact = max(1, prof["k_ldl_diag"][1] // max(1, (9 * ncams + 127) // 128))
        if name == "k_ldl_update":
            # one launch per panel pair (k, k+1), k even: S_ij -= V0_i L_jk' + V1_i L_j,k+1' for the m(m+1)/2 lower tiles
            # right of tile column k+1, m = nt-k-2: 2 tile products of 2*128^3 flop each (kernel table)
            nt = (9 * ncams + 127) // 128
            # N > 1: the factorisation is distributed, a rank updates the tile columns it owns (pairs q with q % N == rank)
            own = [j for j in range(nt) if (j // 2) % world == rank]
            # (single GPU: updates of at most BA_LDL_UPDATE_RS_MAX = 320 tiles take the row-split kernel, its own class)
            rs_max = int(os.environ.get("BA_LDL_UPDATE_RS_MAX", "320")) if world == 1 else 0
            per_launch = [sum(nt - j for j in own if j >= k + 2) for k in range(0, nt - 2, 2)]
            tiles = sum(t for t in per_launch if t > rs_max)
            flops = n_fact * tiles * 2 * 2.0 * 128 ** 3
            ach = flops / (ms * 1e-3) / 1e12
            peak_tf = MFMA_F64_PEAK_TF if args.facto_type == "f64" else MFMA_F32_PEAK_TF
            roof = dict(kernel=name, bound="mfma", achieved=ach, peak=peak_tf, unit="TFLOP/s",
                        frac=ach / peak_tf, traffic=pmc_traffic(args.workload, name),
                        avg_launch_ms=ms / calls, launches=calls, flops_per_launch=flops / calls)
            # the same over ALL pair updates of the factorisation, the short ones in their row-split kernel included -- the
            # definition rounds 1 and 2 quoted (one class then), so that rounds stay comparable
            ms_rs, calls_rs = prof.get("k_ldl_update_rs", (0.0, 0))
            flops_all = n_fact * sum(per_launch) * 2 * 2.0 * 128 ** 3
            ach_all = flops_all / ((ms + ms_rs) * 1e-3) / 1e12
            roof["all_pair_updates"] = dict(kernels="k_ldl_update + k_ldl_update_rs", achieved=ach_all, frac=ach_all / peak_tf,
                                            launches=calls + calls_rs, avg_launch_ms=(ms + ms_rs) / max(1, calls + calls_rs))
    jac_traffic = pmc_traffic(args.workload, "k_jac_coord")  # measured at N = 1; a rank's launch moves its shard's share
    roof_jac = dict(kernel="k_jac_coord", bound="hbm", achieved=jac_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=jac_gbs / HBM_PEAK_GBS, traffic=None if jac_traffic is None else jac_traffic / world,
                    avg_launch_ms=jac_ms,
                    bytes_per_obs=208.0 + 8.0 * nvar_g / nobs_g)
    roof_res = dict(kernel="k_cam_pre + k_residual", bound="hbm", achieved=res_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=res_gbs / HBM_PEAK_GBS, traffic=None, avg_launch_ms=res_ms,
                    bytes_per_obs=48.0 + 8.0 * nvar_g / nobs_g)
    if roof is None:
        roof = roof_jac

    # ---- CPU baseline: the oracle's restatement of the reference LM, bounded sample, rank 0 at N = 1 only ---------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        orc = ge.load_oracle()
        cores = orc.lib().orc_num_threads()
        t0 = time.time()
        rc, _, s, _ = orc.lm_solve(prob["ncams"], prob["npnts"], prob["cam_idx1"], prob["pnt_idx1"], prob["pt2d"],
                                   prob["x0"], variant=1, max_iter_timed=1, facto_time_cap_s=args.cpu_seconds)
        # numeric LDL': the residual-row and point columns (head) are timed exactly; the camera columns (tail) are timed
        # for --cpu-seconds and extrapolated by their multiply-add count
        w_tail_total = s.facto_work_total - s.facto_work_head
        w_tail_done = s.facto_work_done - s.facto_work_head
        t_tail_done = s.t_facto - s.t_facto_head
        frac = w_tail_done / w_tail_total if w_tail_total > 0 else 1.0
        t_fact = s.t_facto_head + (t_tail_done / max(frac, 1e-12) if frac < 1.0 else t_tail_done)
        rate = w_tail_done / max(t_tail_done, 1e-9)
        t_solve = s.t_solve if s.t_solve > 0 else 2.0 * s.lnz / rate  # L and L' sweeps when the solve never ran
        t_sparse = s.t_assemble / s.n_sparse if s.n_sparse else s.t_analyse * 0.0
        # one accepted iteration of the reference = factor + solve + residual + Jacobian + sparse() + J'r
        t_iter = t_fact + t_solve + s.t_residual / max(1, s.n_res) + s.t_jac / max(1, s.n_jac) + s.t_jtr + t_sparse
        cpu = dict(value=1.0 / t_iter, unit="LM iterations/s", cores=cores, kind="port",
                   sample=(f"1 LM iteration of oracle/ba_oracle.c (restated lm.jl + ldl_aux.jl, structured elimination order) "
                           f"on the full {args.workload} shape: residual/Jacobian/J'r and the residual-row + point columns of the "
                           f"numeric LDL' ({s.t_facto_head:.1f}s) timed exactly, its camera columns stopped after {t_tail_done:.1f}s = "
                           f"{100 * frac:.3g}% of their multiply-adds and extrapolated by multiply-add count (symbolic analysis "
                           f"{s.t_analyse:.0f}s, one-off, excluded); residual on {cores} threads, Jacobian on {min(3, cores)}, "
                           f"factorisation on 1"),
                   jacobian_mnnz_per_s=24.0 * nobs_g / (s.t_jac / max(1, s.n_jac)) / 1e6,
                   t_factor_s=t_fact, factor_fraction_timed=frac, lnz=int(s.lnz), wall_s=time.time() - t0)

    # ---- one FULLY timed CPU iteration (no cap, no extrapolation) on a shape the oracle finishes, beside the GPU's time
    # on that same shape from this same process ------------------------------------------------------------------------
    cpu_full = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and args.cpu_full != "none":
        orc = ge.load_oracle()
        prob2 = ba.synthetic.make_named(args.cpu_full)
        nlp2 = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(prob2), device=local_rank, model_name=args.cpu_full)
        fr2 = ba.FeasibilityResidual(nlp2)
        lm_fixed_iterations(ba, fr2, 3)
        torch.cuda.synchronize()
        k2 = 20
        tg = time.perf_counter()
        st2 = lm_fixed_iterations(ba, fr2, k2)
        torch.cuda.synchronize()
        gpu_ms = 1e3 * (time.perf_counter() - tg) / k2
        nlp2.close()
        t0 = time.time()
        rc, _, s2, _ = orc.lm_solve(prob2["ncams"], prob2["npnts"], prob2["cam_idx1"], prob2["pnt_idx1"], prob2["pt2d"],
                                    prob2["x0"], variant=1, max_iter_timed=1)
        # one accepted iteration of lm.jl: factor + solve + trial residual + Jacobian + sparse() + J'r, every phase timed whole
        t_iter = (s2.t_facto / max(1, s2.n_facto) + s2.t_solve / max(1, s2.n_facto) + s2.t_residual / max(1, s2.n_res)
                  + s2.t_jac / max(1, s2.n_jac) + s2.t_jtr / max(1, s2.n_jac) + s2.t_assemble / max(1, s2.n_sparse))
        cores2 = orc.lib().orc_num_threads()
        cpu_full = dict(value=1.0 / t_iter, unit="LM iterations/s", cores=cores2, kind="port", workload=args.cpu_full,
                        sample=(f"1 complete LM iteration of oracle/ba_oracle.c on the {args.cpu_full} shape (ncams={prob2['ncams']} "
                                f"npnts={prob2['npnts']} nobs={prob2['nobs']}): numeric LDL' {s2.t_facto:.2f}s on 1 thread, timed to "
                                f"the end; solve {s2.t_solve:.2f}s; residual on {cores2} threads, Jacobian on {min(3, cores2)}; "
                                f"symbolic analysis {s2.t_analyse:.1f}s (one-off) excluded"),
                        s_per_iteration=t_iter, t_factor_s=s2.t_facto / max(1, s2.n_facto), factor_fraction_timed=1.0,
                        lnz=int(s2.lnz), wall_s=time.time() - t0, rc=rc,
                        gpu_ms_per_step_same_shape=gpu_ms, gpu_it_per_s_same_shape=1e3 / gpu_ms,
                        gpu_over_cpu=(1e3 / gpu_ms) * t_iter,
                        reference_log=("benchmark/first/lm_big.log:158-221 (real Dubrovnik-356 file, unknown CPU): 65 s per "
                                       "iteration" if args.cpu_full == "dubrovnik-356" else None))

    if rank == 0:
        out = {
            "metric": f"LM iterations/sec on BAL {args.workload} (synthetic BAL-shaped), plus Jacobian Mnnz/sec",
            "value": args.steps / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.facto_type == "f64" else "f64 (reduced camera system factored in f32)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} shape: ncams={ncams} npnts={npnts_g} nobs={nobs_g}, seed "
                                   f"{ba.synthetic.BASE_SEED}, lm.jl variant, {FACTO}/None" + (f" (pcg_tol {PCG_TOL:g})" if FACTO == "PCG" else "") + f", facto_type {args.facto_type}, fixed {args.steps} iterations"
                                   + ("" if args.scale == 1.0 else f" SCALED x{args.scale} (debug)")
                                   + ("" if args.locality is None else f" LOCALITY {args.locality} (block-banded S; not the headline configuration)")
                                   + ("" if args.plane_radius is None else f" CAMERAS IN THE PLANE, radius {args.plane_radius} (geometric camera graph, numbering without structure; not the headline configuration)")
                                   + (" CAMERAS RENUMBERED AT RANDOM" if args.shuffle_cameras else "") + f", perm {PERM}"
                                   + (f" EMULATED SHARD {args.emulate_shard} (one rank's compute, no communicator)" if args.emulate_shard else ""),
                       "parallelism": f"points sharded over {world} rank(s), cameras replicated"
                                      + ("" if world == 1 else "; reduced camera matrix reduced onto the owners of its tile column "
                                         "pairs, factorisation distributed (panel broadcast), solves replicated")},
            "jacobian_mnnz_per_s": jac_mnnz,
            "jacobian_ms": jac_ms,
            "host_boundary": {"value": args.steps / elapsed_host, "ms_per_step": 1e3 * elapsed_host / args.steps,
                              "note": "the same K iterations through ba_lm_solve: x0 uploaded from and the solution downloaded to "
                                      "pageable host memory inside the call (PCIe-inclusive; never `value`)"},
            "lm": {"accepted": st.n_accepted, "rejected": st.n_rejected, "objective": st.objective,
                   "n_jacobian": st.n_jacobian, "n_factor": st.n_factor, "n_cg": st.n_cg, "loop_s": st.loop_time},
            "pcg": pcg_line,
            "roofline": roof,
            "roofline_traffic_source": pmc_stamp(args.workload),
            "roofline_jacobian": roof_jac,
            "roofline_residual": roof_res,
            "cpu_baseline": cpu,
            "cpu_baseline_full": cpu_full,
            "kernel_ms": {k: round(v[0], 3) for k, v in prof.items() if v[1] > 0},
            "setup_s": t_setup,
        }
        if FACTO != "PCG":
            tf, ff, sp = ba.schur_pattern(nlp)
            full, held, staging = ba.schur_memory(nlp)
            out["schur_pattern"] = {"camera_sequence": ba.schur_ordering_used(nlp)[1], "tile_fill": tf, "update_tiles_over_dense": ff, "list_schedule": sp,
                                    "tiles_full": full, "tiles_held": held, "tiles_staging": staging,
                                    "S_gib_held": held * 128 * 128 * 8 / 2 ** 30}
        if reducer is not None:
            out["comm"] = {"transport": "rccl (called from the library)" if args.backend == "nccl" else "hook over " + args.backend,
                           "calls": reducer.calls, "bytes": reducer.bytes, "by_operation": reducer.stats_by_op(),
                           "ms_profiled_run": prof.get("allreduce", (None, 0))[0]}
        print(json.dumps(out))
    nlp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
