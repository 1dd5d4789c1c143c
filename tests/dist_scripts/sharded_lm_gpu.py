"""Run by torch.distributed.run with 2 ranks (gloo), both on cuda:0: sharded LM on the real kernels; rank 0 compares
with the unsharded run and writes the verdict to argv[1]."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch, torch.distributed as dist

def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ba = ge.load_package()
    out = dict(rank=rank, world=world)
    # (a) the distributed factorisation on a system with several tile column pairs (70 cameras: n = 630, 5 tile rows,
    # pairs owned 0,1,0 / 0,1,2): one LM step of the shards against the step of the unsharded problem
    big = ba.synthetic.make_problem(70, 500, 2400, seed=8)
    barr = ba.synthetic.as_arrays(big)
    bl, binfo = ba.parallel.shard_problem(barr, rank, world)
    bn = ba.BALNLPModel(arrays=bl, device=0)
    bred = ba.parallel.CameraBlockReducer(bn)
    steps = {}
    for tag, ft in (("f64", None), ("f32", np.float32)):
        d, half, jtr = ba.lm_step(bn, bl[3], 25.0, facto_type=ft)
        steps[tag] = (ba.parallel.gather_solution(d, binfo, big["ncams"]), half, ba.parallel.gather_solution(jtr, binfo, big["ncams"]))
    out["step_calls"] = bred.calls
    bn.close()
    if rank == 0:
        bf = ba.BALNLPModel(arrays=barr, device=0)
        for tag, ft in (("f64", None), ("f32", np.float32)):
            d_ref, half_ref, jtr_ref = ba.lm_step(bf, barr[3], 25.0, facto_type=ft)
            d, half, jtr = steps[tag]
            out["step_" + tag] = float(np.linalg.norm(d - d_ref) / np.linalg.norm(d_ref))
            out["half_" + tag] = float(abs(half - half_ref) / half_ref)
            out["jtr_" + tag] = float(np.max(np.abs(jtr - jtr_ref)) / np.max(np.abs(jtr_ref)))
        bf.close()
    # (a2) 8 tile column pairs (200 cameras: n = 1800, 15 tile rows): the look-ahead's two-deep panel buffers are reused
    # several times; the gathered step travels to the test, which compares the look-ahead run with the alternating one
    wide = ba.synthetic.make_problem(200, 1500, 9000, seed=11)
    warr = ba.synthetic.as_arrays(wide)
    wl, winfo = ba.parallel.shard_problem(warr, rank, world)
    wn = ba.BALNLPModel(arrays=wl, device=0)
    wred = ba.parallel.CameraBlockReducer(wn)  # noqa: F841 (the model keeps it alive too)
    dw, halfw, _ = ba.lm_step(wn, wl[3], 10.0)
    dw = ba.parallel.gather_solution(dw, winfo, wide["ncams"])
    # facto = :PCG on the shards: the product S v is all-reduced once per CG iteration, everything else is replicated
    dpcg, halfpcg, _, its = ba.lm_step(wn, wl[3], 10.0, pcg=(1e-12, 5000))
    dpcg = ba.parallel.gather_solution(dpcg, winfo, wide["ncams"])
    stp = ba.Levenberg_Marquardt(ba.FeasibilityResidual(wn), "PCG", "AMD", "None", False, ite_max=3, pcg_tol=1e-10)
    out.update(pcg_its=its, pcg_lm_objective=stp.objective, pcg_lm_cg=stp.n_cg)
    wn.close()
    if rank == 0:
        wf = ba.BALNLPModel(arrays=warr, device=0)
        dw_ref, halfw_ref, _ = ba.lm_step(wf, warr[3], 10.0)
        stp1 = ba.Levenberg_Marquardt(ba.FeasibilityResidual(wf), "PCG", "AMD", "None", False, ite_max=3, pcg_tol=1e-10)
        out["pcg_lm_objective_ref"] = stp1.objective
        wf.close()
        out["step_wide"] = float(np.linalg.norm(dw - dw_ref) / np.linalg.norm(dw_ref))
        out["half_wide"] = float(abs(halfw - halfw_ref) / halfw_ref)
        out["wide_hex"] = [float(v).hex() for v in dw[-9 * wide["ncams"]:]]
        import hashlib
        out["wide_digest"] = hashlib.sha1(np.ascontiguousarray(dw[-9 * wide["ncams"]:]).tobytes()).hexdigest()[:16]
        out["step_pcg"] = float(np.linalg.norm(dpcg - dw_ref) / np.linalg.norm(dw_ref))
        out["half_pcg"] = float(abs(halfpcg - halfw_ref) / halfw_ref)
    # (b) complete LM runs
    prob = ba.synthetic.make_problem(14, 600, 2700, seed=5)
    arrays = ba.synthetic.as_arrays(prob)
    local, info = ba.parallel.shard_problem(arrays, rank, world)
    nlp = ba.BALNLPModel(arrays=local, device=0)
    red = ba.parallel.CameraBlockReducer(nlp)
    st = ba.Levenberg_Marquardt(ba.FeasibilityResidual(nlp), "LDL", "AMD", "None", False)
    x = ba.parallel.gather_solution(st.solution, info, prob["ncams"])
    out.update(iter=st.iter, objective=st.objective, status=st.status, calls=red.calls)
    # column scaling needs the GLOBAL column norms of J: diag(Hcc) is all-reduced with gc
    stj = ba.Levenberg_Marquardt(ba.FeasibilityResidual(nlp), "LDL", "AMD", "J", False)
    out.update(iter_j=stj.iter, objective_j=stj.objective, status_j=stj.status)
    # a Float32 model (eltype(x) = Float32) on two ranks; tolerances of the reference's Float32 experiment
    tol32 = dict(oatol=1e-4, ortol=1e-4, atol=1e-4, rtol=1e-5, satol=1e-6, srtol=1e-7)
    l32 = list(local)
    l32[2], l32[3] = l32[2].astype(np.float32), l32[3].astype(np.float32)
    nlp32 = ba.BALNLPModel(arrays=tuple(l32), T=np.float32, device=0)
    red32 = ba.parallel.CameraBlockReducer(nlp32)
    st32 = ba.Levenberg_Marquardt(ba.FeasibilityResidual(nlp32), "LDL", "AMD", "None", False, **tol32)
    out.update(objective_32=st32.objective, status_32=st32.status)
    nlp32.close()
    if rank == 0:
        full = ba.BALNLPModel(arrays=arrays, device=0)
        ref = ba.Levenberg_Marquardt(ba.FeasibilityResidual(full), "LDL", "AMD", "None", False)
        out.update(ref_iter=ref.iter, ref_objective=ref.objective, ref_status=ref.status,
                   dx=float(np.linalg.norm(x - ref.solution) / np.linalg.norm(ref.solution)),
                   log_equal=[r[7] for r in st.log] == [r[7] for r in ref.log])
        refj = ba.Levenberg_Marquardt(ba.FeasibilityResidual(full), "LDL", "AMD", "J", False)
        out.update(ref_iter_j=refj.iter, ref_objective_j=refj.objective, ref_status_j=refj.status)
        a32 = list(arrays)
        a32[2], a32[3] = a32[2].astype(np.float32), a32[3].astype(np.float32)
        full32 = ba.BALNLPModel(arrays=tuple(a32), T=np.float32, device=0)
        ref32 = ba.Levenberg_Marquardt(ba.FeasibilityResidual(full32), "LDL", "AMD", "None", False, **tol32)
        out.update(ref_objective_32=ref32.objective, ref_status_32=ref32.status)
        full32.close()
        json.dump(out, open(sys.argv[1], "w"))
        full.close()
    nlp.close()
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
