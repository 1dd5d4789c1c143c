"""Generates the committed golden vectors under tests/golden/ (run once in the build container, where
/root/reference is mounted; the GPU box never runs this).

1. residual_scipy.npz -- inputs of a seeded synthetic problem and the residuals computed by the reference's own
   Python restatement of the camera model: the function definitions of src/SolverScipy.py (lines 1-98: read_bal_data,
   rotate, project, fun -- the rest of that file is a script that needs BAL data files and is not executed).  The
   definitions are exec'ed from the mounted reference at generation time; nothing of the file is copied here.
2. runtests_fixture.npz -- the 5-observation fixture and the known answers that test/runtests.jl:5-27 holds
   (data: inputs and expected outputs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF = "/root/reference/src/SolverScipy.py"


def reference_python_functions():
    src = "".join(open(REF).readlines()[:98])
    ns = {}
    exec(compile(src, REF, "exec"), ns)
    return ns


def main():
    ba = ge.load_package()
    ns = reference_python_functions()
    prob = ba.synthetic.make_problem(7, 90, 400, seed=20261004)
    ncams, npnts = prob["ncams"], prob["npnts"]
    for tag, x in (("x0", prob["x0"]), ("xtrue", prob["x_true"])):
        pts = x[: 3 * npnts].reshape(npnts, 3)
        cams = x[3 * npnts:].reshape(ncams, 9)
        cams_py = cams[:, [0, 1, 2, 3, 4, 5, 8, 6, 7]]  # reference layout (r,t,k1,k2,f) -> BAL/scipy layout (r,t,f,k1,k2)
        params = np.hstack([cams_py.ravel(), pts.ravel()])
        res = ns["fun"](params, ncams, npnts, prob["cam_idx1"] - 1, prob["pnt_idx1"] - 1, prob["pt2d"].reshape(-1, 2))
        prob["res_" + tag] = res
    np.savez_compressed(os.path.join(HERE, "residual_scipy.npz"), cam_idx1=prob["cam_idx1"], pnt_idx1=prob["pnt_idx1"],
                        pt2d=prob["pt2d"], x0=prob["x0"], x_true=prob["x_true"], ncams=ncams, npnts=npnts,
                        res_x0=prob["res_x0"], res_xtrue=prob["res_xtrue"])
    # test/runtests.jl:15-26 (numbers as printed there)
    pt2d = np.array([-3.326500e+02, 2.620900e+02, -1.997600e+02, 1.667000e+02, -2.530600e+02, 2.022700e+02,
                     5.813000e+01, 2.718900e+02, 2.382200e+02, 2.373700e+02])
    x = np.array([-0.6120001571722636, 0.5717590477602829, -1.8470812764548823, 0.01574151594294026,
                  -0.012790936163850642, -0.004400849808198079, -0.034093839577186584, -0.10751387104921525,
                  1.1202240291236032, -3.177064385280358e-7, 5.882049053459402e-13, 399.75152639358436,
                  0.01597732412020533, -0.02522446458285646, -0.00940014164793023, -0.00856676614082241,
                  -0.12188049069425422, 0.719013307500946, -3.7804765613385677e-7, 9.30743116838448e-13,
                  402.0175338595593, 0.014846251175275622, -0.021062899405576294, -0.0011669480098224182,
                  -0.024950970734443037, -0.11398470545726247, 0.9216602073702798, -3.2952646187978145e-7,
                  6.732885068879348e-13, 400.4017536835857, 0.01991666998444233, -1.2243308199651954,
                  0.011998875602428538, -1.411897512312013, -0.11480651507716103, 0.44915582738113896,
                  5.958750036132224e-8, -2.4839062920074967e-13, 407.0302456821108, 0.02082242153136291,
                  -1.238434791463721, 0.013893147632321344, -1.0496862247709429, -0.12995132856190453,
                  0.3379838023131856, 4.5673126640998776e-8, -1.7924276184384984e-13, 405.9176496201471])
    true_residuals = np.array([-9.020226301243156, 11.263958304987227, -1.833229714946924, 5.304698960898122,
                               -4.332321480806684, 7.117305031392988, -0.5632751791502884, -1.062178017695942,
                               -3.96920595468427, -2.285071283095334])
    np.savez(os.path.join(HERE, "runtests_fixture.npz"), pt2d=pt2d, x=x, cam_idx=np.arange(1, 6), pnt_idx=np.ones(5, dtype=np.int64),
             nobs=5, npnts=1, true_residuals=true_residuals,
             # test/runtests.jl:6-8
             rodrigues_r=np.array([1.0, 1.0, 1.0]), rodrigues_x=np.array([2.5, -0.3, 1.0]),
             rodrigues_out=np.array([1.577353756980212, 2.1408840848258484, -0.5182378418060594]),
             scaling_point=np.array([1.0, 1.0]), scaling_k=np.array([1.0, 1.0]), scaling_out=7.0,
             projection_args=np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0]),
             projection_out=np.array([-7.0, -7.0]))
    orc = ge.load_oracle()
    d = orc.residuals(prob["cam_idx1"], prob["pnt_idx1"], prob["x0"], prob["pt2d"], npnts) - prob["res_x0"]
    print("oracle vs reference-python residuals, max |diff| =", np.abs(d).max())


if __name__ == "__main__":
    main()
