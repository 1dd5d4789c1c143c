"""N > 1 path.  CPU: the ownership map of the distributed reduced camera matrix; (gloo, world_size 2) sharding helpers +
the cross-rank sums of the design, emulated with the oracle.  GPU: (i) 2 and 3 gloo ranks sharing cuda:0 run the sharded
LM with the distributed factorisation on the real kernels (hook transport) and must reproduce the single-rank run;
(ii) a one-rank RCCL communicator opened by the library itself drives every collective from C."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(script, nproc, tmp_path, timeout=600, env=None):
    out = str(tmp_path / "out.json")
    if os.path.exists(out):
        os.remove(out)
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_scripts", script), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


def test_sharded_sums_gloo_cpu(tmp_path):
    res = _run_ranks("sharded_sums_cpu.py", 2, tmp_path)
    assert res["rc"] == 0
    assert res["S"] < 1e-13 and res["rhs"] < 1e-12 and res["gc"] < 1e-13
    assert res["delta"] < 1e-9


def test_dist_layout_ownership_map(ba):
    """Layout of the reduced camera matrix over the ranks (ba_dist_layout, host-only): the tile columns tile the packed
    lower triangle exactly once, a pair (2q, 2q+1) is adjacent and owned by q % world, a rank's columns are contiguous."""
    for nt in (1, 2, 3, 7, 26, 126, 963):
        for world in (1, 2, 3, 8):
            col_off, own = ba.parallel.dist_layout(nt, world)
            assert own[0] == 0 and own[-1] == nt * (nt + 1) // 2
            spans = sorted((int(col_off[j]), int(col_off[j]) + nt - j, j) for j in range(nt))
            assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == own[-1]
            for j in range(nt):
                r = (j // 2) % world
                assert own[r] <= col_off[j] and col_off[j] + nt - j <= own[r + 1]
                if j % 2 == 1:
                    assert col_off[j] == col_off[j - 1] + nt - (j - 1)
            # inside a rank: ascending pair order
            for r in range(world):
                mine = [j for j in range(nt) if (j // 2) % world == r]
                assert [int(col_off[j]) for j in mine] == sorted(int(col_off[j]) for j in mine)


def _check_sharded(res, tag="sharded run"):
    """Every check names its field and value: a truncated log still says what failed."""
    shown = {k: v for k, v in res.items() if k != "wide_hex"}
    print(tag, shown, "wide digest", res.get("wide_digest"))

    def le(key, limit):
        assert res[key] <= limit, f"{tag}: {key} = {res[key]!r} exceeds {limit!r}"

    def same(a, b):
        assert res[a] == res[b], f"{tag}: {a} = {res[a]!r} but {b} = {res[b]!r}"

    def close(a, b, rtol):
        assert abs(res[a] - res[b]) <= rtol * abs(res[b]), f"{tag}: {a} = {res[a]!r} vs {b} = {res[b]!r} (relative limit {rtol:g})"

    # distributed factorisation: the sharded step equals the unsharded one (Float64: rounding of a different summation
    # order only; Float32 factorisation: Float32 level)
    le("step_f64", 1e-9), le("half_f64", 1e-10), le("jtr_f64", 1e-12)
    le("step_f32", 5e-3), le("jtr_f32", 1e-12)
    le("step_wide", 1e-9), le("half_wide", 1e-10)
    # facto = :PCG on the shards: the tightly solved CG step is the direct step; three LM iterations as on one rank
    assert 0 < res["pcg_its"] < 5000, f"{tag}: pcg_its = {res['pcg_its']!r}"
    le("step_pcg", 1e-8), le("half_pcg", 1e-9)
    close("pcg_lm_objective", "pcg_lm_objective_ref", 1e-9)
    assert res["pcg_lm_cg"] > 0 and res["step_calls"] > 0, f"{tag}: pcg_lm_cg = {res['pcg_lm_cg']!r}, step_calls = {res['step_calls']!r}"
    same("iter", "ref_iter"), same("status", "ref_status")
    assert res["log_equal"], f"{tag}: accept / reject sequence differs from the one-rank run"
    close("objective", "ref_objective", 1e-9)
    le("dx", 1e-7)
    assert res["calls"] >= 2 * res["iter"], f"{tag}: calls = {res['calls']!r}, iter = {res['iter']!r}"
    # normalize = :J on several ranks: same run as on one (the scaling uses the all-reduced diagonal of J'J)
    same("iter_j", "ref_iter_j"), same("status_j", "ref_status_j")
    close("objective_j", "ref_objective_j", 1e-9)
    # Float32 iterates: sharding changes the order of the camera-side sums, Float32 rounding of x can amplify that
    close("objective_32", "ref_objective_32", 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_sharded_lm_ranks_share_one_gpu(tmp_path, gpu_ok, nproc):
    """2 and 3 ranks (gloo, hook transport) on cuda:0: observations sharded by point, reduced camera matrix reduced onto
    the owners of its tile column pairs, factorisation distributed, solves replicated -- must reproduce the one-rank run."""
    _check_sharded(_run_ranks("sharded_lm_gpu.py", nproc, tmp_path), f"{nproc} ranks")


@pytest.mark.gpu
def test_dist_lookahead_equals_alternating_schedule(tmp_path, gpu_ok):
    """The distributed factorisation with look-ahead (owner of the next pair updates its leading columns first, runs its
    chain and broadcasts beside the rest of the update) does the same arithmetic as the strictly alternating schedule
    (BA_DIST_LOOKAHEAD=0): the camera part of the step is bit-identical on 3 ranks (gloo hook transport; the same
    comparison with asynchronous streams in one process: tests/test_gpu_determinism.py)."""
    from _util import bits_report
    a = _run_ranks("sharded_lm_gpu.py", 3, tmp_path)
    _check_sharded(a, "3 ranks, look-ahead")
    b = _run_ranks("sharded_lm_gpu.py", 3, tmp_path, env={"BA_DIST_LOOKAHEAD": "0"})
    _check_sharded(b, "3 ranks, alternating")
    wa = np.array([float.fromhex(h) for h in a["wide_hex"]])
    wb = np.array([float.fromhex(h) for h in b["wide_hex"]])
    rep = bits_report(wa, wb, "camera step of the 200-camera problem, look-ahead vs alternating")
    assert not rep, rep


@pytest.mark.gpu
def test_sharded_lm_replicated_factor_switch(tmp_path, gpu_ok):
    """BA_DIST_FACTOR=0: one all-reduce of S and a replicated factorisation (round 1's scheme) stays available."""
    _check_sharded(_run_ranks("sharded_lm_gpu.py", 2, tmp_path, env={"BA_DIST_FACTOR": "0"}), "2 ranks, replicated factor")


@pytest.mark.gpu
def test_rccl_single_rank(ba, small_prob, gpu_ok):
    """A one-rank RCCL communicator opened BY THE LIBRARY (ba_comm_get_unique_id / ba_lm_set_comm_rccl; torch.distributed
    only ferries the 128-byte id): every collective of the multi-GPU path (all-reduce, reduce onto the owner, panel
    broadcast, grouped calls) runs through RCCL from C on the library's stream.  Against the plain single-GPU run: same
    iterations; the objective differs by rounding only (the distributed schedule does the forward substitution as its own
    sweep instead of fused into the panel solves)."""
    import torch
    import torch.distributed as dist
    p = small_prob
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
    st_ref = ba.Levenberg_Marquardt(ba.FeasibilityResidual(ref), "LDL", "AMD", "None", False)
    big = ba.synthetic.make_problem(70, 500, 2400, seed=8)
    bref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(big))
    d_ref, half_ref, _ = ba.lm_step(bref, big["x0"], 25.0)
    d32_ref, _, _ = ba.lm_step(bref, big["x0"], 25.0, facto_type=np.float32)
    bref.close()
    wide = ba.synthetic.make_problem(200, 1500, 9000, seed=11)  # 15 tile rows: 8 pairs through the look-ahead's buffers
    wref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(wide))
    dw_ref, halfw_ref, _ = ba.lm_step(wref, wide["x0"], 10.0)
    wref.close()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")  # initialise torch's HIP context before c10d counts the devices
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        m = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
        red = ba.parallel.CameraBlockReducer(m)
        st = ba.Levenberg_Marquardt(ba.FeasibilityResidual(m), "LDL", "AMD", "None", False)
        assert red.calls >= 2 * st.iter and red.bytes > 0
        assert st.iter == st_ref.iter and st.status == st_ref.status
        assert abs(st.objective - st_ref.objective) <= 1e-10 * st_ref.objective
        assert np.linalg.norm(st.solution - st_ref.solution) <= 1e-8 * np.linalg.norm(st_ref.solution)
        m.close()
        bm = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(big))
        ba.parallel.CameraBlockReducer(bm)
        d, half, _ = ba.lm_step(bm, big["x0"], 25.0)
        assert np.linalg.norm(d - d_ref) <= 1e-10 * np.linalg.norm(d_ref) and abs(half - half_ref) <= 1e-11 * half_ref
        # facto_type = Float32: the partial sums of the reduced camera matrix travel as Float32 (ncclFloat32 reduce)
        d32, _, _ = ba.lm_step(bm, big["x0"], 25.0, facto_type=np.float32)
        assert np.linalg.norm(d32 - d32_ref) <= 1e-4 * np.linalg.norm(d32_ref)
        bm.close()
        wm = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(wide))
        ba.parallel.CameraBlockReducer(wm)
        for _ in range(2):  # twice: the second factorisation re-uses events and panel buffers of the first
            dw, halfw, _ = ba.lm_step(wm, wide["x0"], 10.0)
            assert np.linalg.norm(dw - dw_ref) <= 1e-10 * np.linalg.norm(dw_ref) and abs(halfw - halfw_ref) <= 1e-11 * halfw_ref
        wm.close()
    finally:
        dist.destroy_process_group()
    ref.close()


@pytest.mark.gpu
def test_bench_two_ranks_launch_path(gpu_ok):
    """`python bench.py --gpus 2` exactly as the driver starts an N > 1 run when no launcher is around it: the parent spawns
    its ranks (torch.distributed.run, 127.0.0.1) BEFORE anything touches the GPU, the ranks shard the problem, one JSON line
    comes back from rank 0.  Two gloo ranks on the one GPU of this box (--single-device --backend gloo), a scaled-down Venice
    shape; the objective after the fixed iterations equals the one-rank run's."""
    def run(extra):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "0.05", "--steps", "3", "--warmup", "1",
               "--cpu-seconds", "0", "--cpu-full", "none", "--no-pcg"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, f"bench.py {' '.join(extra)} failed:\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}: {r.stdout[-1500:]}"
        return json.loads(lines[0])

    one = run([])
    two = run(["--gpus", "2", "--single-device", "--backend", "gloo"])
    assert one["n_gpus"] == 1 and "comm" not in one
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["warmup"] == 1 and two["scaling"] == "strong"
    assert two["value"] > 0 and abs(two["value"] * two["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert "comm" in two and two["comm"]["calls"] > 0 and two["comm"]["bytes"] > 0, two.get("comm")
    assert two["roofline"]["frac"] > 0 and two["cpu_baseline"] is None
    f1, f2 = one["lm"]["objective"], two["lm"]["objective"]
    assert abs(f2 - f1) <= 1e-9 * f1, f"objective after 3 iterations: 2 ranks {f2!r} vs 1 rank {f1!r}"
