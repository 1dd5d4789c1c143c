"""Command-line driver with the positional arguments of the reference's src/solve_ba.jl:4-27:

    python -m bundleadjustment_jl_amd.solve_ba <file> <QR|LDL> <AMD|Metis> <None|A|J>

(run it as `python "bundleadjustment.jl_amd/solve_ba.py" ...` from the repository root).  `file` is resolved like
src/ReadFiles.jl:10 does: relative to <repo>/Data unless it exists as given.
"""
import os
import sys


def main(argv):
    if len(argv) != 4:
        print(__doc__)
        return 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ba = ge.load_package()
    facto = {"QR": "QR", "LDL": "LDL"}[argv[1]]          # solve_ba.jl:4-8
    perm = {"AMD": "AMD", "Metis": "Metis"}[argv[2]]     # :10-14
    norm = {"None": "None", "A": "A", "J": "J"}[argv[3]]  # :16-22
    BA = ba.BALNLPModel(argv[0])                         # :24
    fr_BA = ba.FeasibilityResidual(BA)                   # :25
    stats = ba.Levenberg_Marquardt(fr_BA, facto, perm, norm, verbose=True)  # :26
    print("\n ------------ \nStats : \n", stats)         # :27
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
