// Probe: accuracy of v_rcp_f64 on this chip and of one / two Newton steps on top of it (the pivot chain of the diagonal-tile
// kernel pays for every dependent FMA).  Prints the largest relative errors against 1/d computed by IEEE division.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *d, double *e0, double *e1, double *e2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = d[i], ref = 1.0 / v;
  double x0 = __builtin_amdgcn_rcp(v);
  double x1 = __builtin_fma(__builtin_fma(-v, x0, 1.0), x0, x0);
  double x2 = __builtin_fma(__builtin_fma(-v, x1, 1.0), x1, x1);
  e0[i] = fabs(x0 - ref) / fabs(ref);
  e1[i] = fabs(x1 - ref) / fabs(ref);
  e2[i] = fabs(x2 - ref) / fabs(ref);
}
int main() {
  const int n = 1 << 22;
  double *h = new double[n], *d, *e[3];
  unsigned long long z = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    z ^= z << 13; z ^= z >> 7; z ^= z << 17;
    double m = 1.0 + (double)(z >> 11) / 9007199254740992.0;          // mantissa in [1, 2)
    int ex = (int)((z >> 3) % 600) - 300;
    h[i] = ldexp(m, ex) * ((z & 1) ? 1 : -1);
  }
  hipMalloc(&d, n * sizeof(double));
  for (auto &p : e) hipMalloc(&p, n * sizeof(double));
  hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, e[0], e[1], e[2], n);
  hipDeviceSynchronize();
  for (int q = 0; q < 3; q++) {
    hipMemcpy(h, e[q], n * sizeof(double), hipMemcpyDeviceToHost);
    double mx = 0; int nz = 0;
    for (int i = 0; i < n; i++) { if (h[i] > mx) mx = h[i]; if (h[i] != 0) nz++; }
    printf("%d Newton step(s): max relative error %.3e (%.2f ulp of 2^-53), %d of %d inexact\n", q, mx, mx / 1.1102230246251565e-16, nz, n);
  }
  return 0;
}
