"""Dense LDL' micro-benchmark on the GPU box: factor time / TFLOP/s at Venice size (n = 16002) + residual check."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ba = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16002
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
rng = np.random.default_rng(0)
R = rng.standard_normal((n, n)).astype(np.float64)
A = R + R.T
A[np.diag_indices(n)] += 4.0 * np.sqrt(n)
del R
b = rng.standard_normal(n)
for _ in range(reps):
    t = time.time()
    x, ms = ba._lib.dense_ldl_solve(A, b, f32=f32)
    print(f"n={n} factor {ms:.2f} ms = {n**3/3/ms/1e9:.2f} TFLOP/s (wall {time.time()-t:.1f}s)", flush=True)
res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
print("relative residual", res)
assert res < (1e-3 if f32 else 1e-10)
