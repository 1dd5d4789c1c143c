"""N > 1 path.  CPU (gloo, world_size 2): sharding helpers + the cross-rank sums of the design, emulated with the
oracle.  GPU: (i) two gloo ranks sharing cuda:0 run the sharded LM on the real kernels and must reproduce the
single-rank run; (ii) a one-rank NCCL group drives the all-reduce hook on the library's stream."""
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


def _run_ranks(script, nproc, tmp_path, timeout=600):
    out = str(tmp_path / "out.json")
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
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


@pytest.mark.gpu
def test_sharded_lm_two_ranks_one_gpu(tmp_path, gpu_ok):
    res = _run_ranks("sharded_lm_gpu.py", 2, tmp_path)
    print(res)
    assert res["iter"] == res["ref_iter"] and res["status"] == res["ref_status"] and res["log_equal"]
    assert abs(res["objective"] - res["ref_objective"]) <= 1e-9 * res["ref_objective"]
    assert res["dx"] <= 1e-7
    assert res["calls"] >= 2 * res["iter"]
    # normalize = :J on two ranks: same run as on one (the scaling uses the all-reduced diagonal of J'J)
    assert res["iter_j"] == res["ref_iter_j"] and res["status_j"] == res["ref_status_j"]
    assert abs(res["objective_j"] - res["ref_objective_j"]) <= 1e-9 * res["ref_objective_j"]
    # Float32 iterates: sharding changes the order of the camera-side sums, Float32 rounding of x can amplify that
    assert abs(res["objective_32"] - res["ref_objective_32"]) <= 1e-3 * res["ref_objective_32"]


@pytest.mark.gpu
def test_allreduce_hook_nccl_single_rank(ba, small_prob, gpu_ok):
    import torch
    import torch.distributed as dist
    p = small_prob
    ref = ba.BALNLPModel(arrays=ba.synthetic.as_arrays(p))
    st_ref = ba.Levenberg_Marquardt(ba.FeasibilityResidual(ref), "LDL", "AMD", "None", False)
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
        assert st.iter == st_ref.iter and st.objective == st_ref.objective  # one rank: bit-identical
        assert np.array_equal(st.solution, st_ref.solution)
        m.close()
    finally:
        dist.destroy_process_group()
    ref.close()
