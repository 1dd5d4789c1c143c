"""Soak test of the hoisted-diagonal factorisation: many factor/solve calls at Venice size, every solution checked.
python tools/soak_ldl.py [n=16002] [reps=100]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ba = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16002
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(0)
R = rng.standard_normal((n, n))
A = R + R.T
A[np.diag_indices(n)] += 4.0 * np.sqrt(n)
del R
worst, t0, ms_all = 0.0, time.time(), []
for r in range(reps):
    b = rng.standard_normal(n)
    x, ms = ba._lib.dense_ldl_solve(A, b)
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    worst = max(worst, res)
    ms_all.append(ms)
    assert res < 1e-10, (r, res)
    if r % 20 == 0:
        print(f"rep {r}: factor {ms:.2f} ms, residual {res:.2e}", flush=True)
print(f"{reps} factorisations ok, worst residual {worst:.2e}, factor ms min/median/max "
      f"{min(ms_all):.2f}/{sorted(ms_all)[len(ms_all) // 2]:.2f}/{max(ms_all):.2f}, wall {time.time() - t0:.0f}s")
