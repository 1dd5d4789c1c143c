// Dense blocked LDL' (no pivoting) of the reduced camera system on the gfx950 f64 matrix cores.
//
// Replaces the camera tail of the reference's scalar up-looking sparse LDL' (src/ldl_aux.jl:122-201, which is
// where 64-97 % of every reference iteration goes) and its triangular solves (src/ldl_aux.jl:4-42).  Like the
// reference it is an LDL' without pivoting that only fails on an exactly zero pivot (SQDException,
// src/ldl_aux.jl:199 -> BA_ERR_ZERO_PIVOT); S is symmetric positive definite in exact arithmetic.
//
// Storage: lower block triangle of NB x NB (128) tiles, each tile contiguous row-major (ba_internal.h).
// Right-looking, one panel (tile column) per step k:
//   k_ldl_diag : one workgroup factors tile (k,k) in LDS (L_kk, D_k) and forms L_kk^-1 explicitly;
//   k_ldl_trsm : X_i = S_ik L_kk^-T  (= L_ik D_k) as an MFMA GEMM with L_kk^-1, stores V_i = X_i and L_ik = X_i D_k^-1;
//   k_ldl_syrk : S_ij -= V_i L_jk' for k < j <= i, MFMA GEMM (v_mfma_f64_16x16x4_f64), the n^3/3 flops.
// Solves: forward sweep by tile columns, diagonal scaling folded into the backward sweep by tile rows.
#include "ba_internal.h"

#include <cmath>
#include <cstdio>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int LDA = NB + 1;  // LDS row stride of the diagonal-tile kernel (bank-conflict padding)
constexpr int KC = 32;       // K chunk of the GEMM kernels staged through LDS
constexpr int LDK = KC + 2;  // row stride 68 dwords: 4i+2k distinct banks for the MFMA operand reads
constexpr size_t DIAG_LDS = (size_t)NB * LDA * sizeof(double);
constexpr size_t GEMM_LDS = (size_t)2 * NB * LDK * sizeof(double);

// v_mfma_f64_16x16x4_f64 C/D layout: lane l, result register g hold C[row][l & 15]
__device__ inline int mfma_row(int lane, int reg) { return (lane >> 4) + 4 * reg; }

// ---- diagonal tile ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ldl_diag(double *__restrict__ Skk, double *__restrict__ Linv_k,
                                                    double *__restrict__ D_k, int *__restrict__ flag) {
  extern __shared__ double a[];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 1024) {
    int i = idx >> 7, j = idx & (NB - 1);
    a[i * LDA + j] = (j <= i) ? Skk[idx] : 0.0;
  }
  __syncthreads();
  const int row = tid >> 3, sub = tid & 7;
  for (int j = 0; j < NB; j++) {
    const double d = a[j * LDA + j];
    const double inv_d = 1.0 / d;
    if (tid == 0) {
      D_k[j] = d;
      if (d == 0.0) *flag = 1;
    }
    if (row > j) {
      const double lij = a[row * LDA + j] * inv_d;
      for (int c = j + 1 + sub; c <= row; c += 8) a[row * LDA + c] -= lij * a[c * LDA + j];
    }
    __syncthreads();
    if (sub == 0 && row > j) a[row * LDA + j] *= inv_d;  // column j is final; nobody reads it again in this loop
  }
  __syncthreads();
  // X = L^-1 (unit lower): X[m][c], m > c, kept at a[c][m] (upper half).  8 lanes per column, all in one wave.
  {
    const int c = tid >> 3;
    for (int i = c + 1; i < NB; i++) {
      double s = 0;
      for (int m = c + 1 + sub; m < i; m += 8) s += a[i * LDA + m] * a[c * LDA + m];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (sub == 0) a[c * LDA + i] = -(a[i * LDA + c] + s);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NB * NB; idx += 1024) {
    int i = idx >> 7, j = idx & (NB - 1);
    double l, x;
    if (j < i) {
      l = a[i * LDA + j];
      x = a[j * LDA + i];
    } else if (j == i) {
      l = a[i * LDA + i];  // D on the diagonal of the stored tile (informative only)
      x = 1.0;
    } else {
      l = 0.0;
      x = 0.0;
    }
    Skk[idx] = l;
    Linv_k[idx] = x;
  }
}

// ---- 128x128x128 tile product C = A * B' on the matrix cores ------------------------------------------------------
// A, B: contiguous row-major tiles in global memory.  256 threads = 4 waves, wave w owns the 64x64 quadrant
// (w >> 1, w & 1) as 4x4 MFMA blocks.
__device__ inline void tile_gemm_abt(const double *__restrict__ A, const double *__restrict__ B, double *sA, double *sB,
                                     d4 acc[4][4]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  const int fr = lane & 15, fk = lane >> 4;
  for (int k0 = 0; k0 < NB; k0 += KC) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (NB * KC / 2) / 256; it++) {
      int idx = it * 256 + tid;
      int row = idx >> 4, c2 = idx & 15;
      double2 va = *reinterpret_cast<const double2 *>(A + row * NB + k0 + 2 * c2);
      double2 vb = *reinterpret_cast<const double2 *>(B + row * NB + k0 + 2 * c2);
      *reinterpret_cast<double2 *>(sA + row * LDK + 2 * c2) = va;
      *reinterpret_cast<double2 *>(sB + row * LDK + 2 * c2) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC / 4; kk++) {
      double af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; m++) af[m] = sA[(wr + 16 * m + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int n = 0; n < 4; n++) bf[n] = sB[(wc + 16 * n + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
  }
}

// X_i = S_ik * Linv_k'  ->  V_i = X_i,  S_ik = X_i * D_k^-1      (i = k+1+blockIdx.x)
__global__ __launch_bounds__(256) void k_ldl_trsm(double *__restrict__ S, const double *__restrict__ Linv_k,
                                                   const double *__restrict__ D_k, double *__restrict__ V, int k) {
  extern __shared__ double lds[];
  double *sA = lds, *sB = lds + NB * LDK;
  const int i = k + 1 + blockIdx.x;
  double *Sik = S + tile_index(i, k) * NB * NB;
  double *Vi = V + (int64_t)i * NB * NB;
  d4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_abt(Sik, Linv_k, sA, sB, acc);
  __syncthreads();  // every wave has finished reading Sik through LDS staging before it is overwritten
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int col = wc + 16 * n + (lane & 15);
    const double inv_d = 1.0 / D_k[col];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = wr + 16 * m + mfma_row(lane, g);
        const double xv = acc[m][n][g];
        Vi[row * NB + col] = xv;
        Sik[row * NB + col] = xv * inv_d;
      }
  }
}

// S_ij -= V_i * L_jk'   for the lower-triangular tile pairs k < j <= i
__global__ __launch_bounds__(256) void k_ldl_syrk(double *__restrict__ S, const double *__restrict__ V, int k) {
  extern __shared__ double lds[];
  double *sA = lds, *sB = lds + NB * LDK;
  const int t = blockIdx.x;
  int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
  while (ii * (ii + 1) / 2 > t) ii--;
  const int jj = t - ii * (ii + 1) / 2;
  const int i = k + 1 + ii, j = k + 1 + jj;
  const double *Vi = V + (int64_t)i * NB * NB;
  const double *Ljk = S + tile_index(j, k) * NB * NB;
  double *Sij = S + tile_index(i, j) * NB * NB;
  d4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_abt(Vi, Ljk, sA, sB, acc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int col = wc + 16 * n + (lane & 15);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = wr + 16 * m + mfma_row(lane, g);
        Sij[row * NB + col] -= acc[m][n][g];
      }
  }
}

// ---- triangular solves -------------------------------------------------------------------------------------------------
__device__ inline double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// forward step k: y_k = Linv_k b_k (every workgroup recomputes it; block 0 stores it), then b_i -= L_ik y_k, i > k.
__global__ __launch_bounds__(256) void k_fwd_step(const double *__restrict__ S, const double *__restrict__ Linv,
                                                   double *__restrict__ b, double *__restrict__ y, int k) {
  __shared__ double yk[NB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double *Lk = Linv + (int64_t)k * NB * NB;
  const double2 bk = *reinterpret_cast<const double2 *>(b + (int64_t)k * NB + 2 * lane);
  for (int rr = 0; rr < 32; rr++) {
    int row = wv * 32 + rr;
    double2 l = *reinterpret_cast<const double2 *>(Lk + row * NB + 2 * lane);
    double s = wsum(l.x * bk.x + l.y * bk.y);
    if (lane == 0) yk[row] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < NB) y[(int64_t)k * NB + tid] = yk[tid];
    return;
  }
  const int i = k + blockIdx.x;
  const double *Lik = S + tile_index(i, k) * NB * NB;
  const double y0 = yk[2 * lane], y1 = yk[2 * lane + 1];
  for (int rr = 0; rr < 32; rr++) {
    int row = wv * 32 + rr;
    double2 l = *reinterpret_cast<const double2 *>(Lik + row * NB + 2 * lane);
    double s = wsum(l.x * y0 + l.y * y1);
    if (lane == 0) b[(int64_t)i * NB + row] -= s;
  }
}

// backward step k: x_k = Linv_k' z_k with z = y / D (every workgroup recomputes it; block 0 stores it into b_k),
// then y_j -= D_j (L_kj' x_k) ... expressed on z: z_j -= L_kj' x_k, i.e. y_j -= D_j * (L_kj' x_k), j < k.
__global__ __launch_bounds__(256) void k_bwd_step(const double *__restrict__ S, const double *__restrict__ Linv,
                                                   const double *__restrict__ D, double *__restrict__ y,
                                                   double *__restrict__ x, int k) {
  __shared__ double zk[NB], xk[NB], part[2][NB];
  const int tid = threadIdx.x;
  const int c = tid & (NB - 1), half = tid >> 7;
  if (tid < NB) zk[tid] = y[(int64_t)k * NB + tid] / D[(int64_t)k * NB + tid];
  __syncthreads();
  {
    const double *Lk = Linv + (int64_t)k * NB * NB;
    double s = 0;
    for (int r = half * 64; r < half * 64 + 64; r++) s += Lk[r * NB + c] * zk[r];
    part[half][c] = s;
  }
  __syncthreads();
  if (tid < NB) xk[tid] = part[0][tid] + part[1][tid];
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < NB) x[(int64_t)k * NB + tid] = xk[tid];
    return;
  }
  const int j = blockIdx.x - 1;  // 0 .. k-1
  const double *Lkj = S + tile_index(k, j) * NB * NB;
  double s = 0;
  for (int r = half * 64; r < half * 64 + 64; r++) s += Lkj[r * NB + c] * xk[r];
  part[half][c] = s;
  __syncthreads();
  if (tid < NB) y[(int64_t)j * NB + tid] -= D[(int64_t)j * NB + tid] * (part[0][tid] + part[1][tid]);
}

}  // namespace

int64_t dense_ldl_tiles_doubles(int64_t n_unpadded) {
  int64_t nt = (n_unpadded + NB - 1) / NB;
  if (nt < 1) nt = 1;
  return nt * (nt + 1) / 2 * NB * NB;
}

static bool g_attr_done = false;
static int set_kernel_attrs() {
  if (g_attr_done) return BA_OK;
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_diag),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG_LDS));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_trsm),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_syrk),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS));
  g_attr_done = true;
  return BA_OK;
}

int dense_ldl_alloc(DenseLDL *w, int64_t n_unpadded, double *external_S) {
  int64_t nt = (n_unpadded + NB - 1) / NB;
  if (nt < 1) nt = 1;
  w->n = nt * NB;
  w->nt = nt;
  if (external_S) {
    w->S = external_S;
    w->own_S = false;
  } else {
    BA_HIP_CHECK(hipMalloc((void **)&w->S, (size_t)dense_ldl_tiles_doubles(n_unpadded) * sizeof(double)));
    w->own_S = true;
  }
  BA_HIP_CHECK(hipMalloc((void **)&w->V, (size_t)nt * NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&w->Linv, (size_t)nt * NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&w->D, (size_t)nt * NB * 2 * sizeof(double)));  // D | y scratch
  BA_HIP_CHECK(hipMalloc((void **)&w->flag, sizeof(int)));
  return set_kernel_attrs();
}

void dense_ldl_free(DenseLDL *w) {
  if (w->own_S && w->S) (void)hipFree(w->S);
  if (w->V) (void)hipFree(w->V);
  if (w->Linv) (void)hipFree(w->Linv);
  if (w->D) (void)hipFree(w->D);
  if (w->flag) (void)hipFree(w->flag);
  *w = DenseLDL();
}

int dense_ldl_factor(ba_problem *p, DenseLDL *w, hipStream_t st, int *zero_pivot) {
  const int nt = (int)w->nt;
  BA_HIP_CHECK(hipMemsetAsync(w->flag, 0, sizeof(int), st));
  for (int k = 0; k < nt; k++) {
    {
      ProfScope ps(p, PC_LDL_DIAG, st);
      hipLaunchKernelGGL(k_ldl_diag, dim3(1), dim3(1024), DIAG_LDS, st, w->S + tile_index(k, k) * NB * NB,
                         w->Linv + (int64_t)k * NB * NB, w->D + (int64_t)k * NB, w->flag);
    }
    const int m = nt - k - 1;
    if (m > 0) {
      {
        ProfScope ps(p, PC_LDL_TRSM, st);
        hipLaunchKernelGGL(k_ldl_trsm, dim3(m), dim3(256), GEMM_LDS, st, w->S, w->Linv + (int64_t)k * NB * NB,
                           w->D + (int64_t)k * NB, w->V, k);
      }
      {
        ProfScope ps(p, PC_LDL_SYRK, st);
        hipLaunchKernelGGL(k_ldl_syrk, dim3(m * (m + 1) / 2), dim3(256), GEMM_LDS, st, w->S, w->V, k);
      }
    }
  }
  BA_HIP_CHECK(hipGetLastError());
  if (zero_pivot) {
    int h = 0;
    BA_HIP_CHECK(hipMemcpyAsync(&h, w->flag, sizeof(int), hipMemcpyDeviceToHost, st));
    BA_HIP_CHECK(hipStreamSynchronize(st));
    *zero_pivot = h;
  }
  return BA_OK;
}

int dense_ldl_solve(ba_problem *p, DenseLDL *w, double *d_b, hipStream_t st) {
  const int nt = (int)w->nt;
  double *y = w->D + (int64_t)nt * NB;
  ProfScope ps(p, PC_SOLVE, st);
  for (int k = 0; k < nt; k++)
    hipLaunchKernelGGL(k_fwd_step, dim3(nt - k), dim3(256), 0, st, w->S, w->Linv, d_b, y, k);
  for (int k = nt - 1; k >= 0; k--)
    hipLaunchKernelGGL(k_bwd_step, dim3(k + 1), dim3(256), 0, st, w->S, w->Linv, w->D, y, d_b, k);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// ---- C ABI: standalone dense solve (tests, roofline measurement) --------------------------------------------------------
extern "C" int ba_dense_ldl_solve(int device, int64_t n, const double *a_lower_rowmajor, const double *b, double *x,
                                  double *factor_ms) {
  if (n <= 0 || !a_lower_rowmajor || !b || !x) {
    ba_set_error("ba_dense_ldl_solve: bad argument");
    return BA_ERR_ARG;
  }
  BA_HIP_CHECK(hipSetDevice(device));
  ba_problem tmp;  // only used for its (disabled) profiling slots
  DenseLDL w;
  int rc = dense_ldl_alloc(&w, n, nullptr);
  if (rc != BA_OK) return rc;
  const int64_t nt = w.nt, npad = w.n;
  std::vector<double> tiles((size_t)dense_ldl_tiles_doubles(n), 0.0);
  for (int64_t i = 0; i < npad; i++) {
    int64_t ti = i / NB;
    for (int64_t j = 0; j <= i; j++) {
      int64_t tj = j / NB;
      double v = (i < n) ? a_lower_rowmajor[i * n + j] : (i == j ? 1.0 : 0.0);
      tiles[(size_t)((tile_index(ti, tj) * NB + (i - ti * NB)) * NB + (j - tj * NB))] = v;
    }
  }
  std::vector<double> bb((size_t)npad, 0.0);
  for (int64_t i = 0; i < n; i++) bb[(size_t)i] = b[i];
  double *d_b = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipMalloc((void **)&d_b, (size_t)npad * sizeof(double)));
  BA_HIP_CHECK(hipMemcpy(w.S, tiles.data(), tiles.size() * sizeof(double), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipMemcpy(d_b, bb.data(), (size_t)npad * sizeof(double), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  int zp = 0;
  BA_HIP_CHECK(hipEventRecord(e0, st));
  rc = dense_ldl_factor(&tmp, &w, st, nullptr);
  BA_HIP_CHECK(hipEventRecord(e1, st));
  if (rc == BA_OK) rc = dense_ldl_solve(&tmp, &w, d_b, st);
  BA_HIP_CHECK(hipMemcpy(&zp, w.flag, sizeof(int), hipMemcpyDeviceToHost));
  BA_HIP_CHECK(hipDeviceSynchronize());
  float ms = 0;
  BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (factor_ms) *factor_ms = ms;
  BA_HIP_CHECK(hipMemcpy(bb.data(), d_b, (size_t)npad * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; i++) x[i] = bb[(size_t)i];
  (void)hipFree(d_b);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dense_ldl_free(&w);
  (void)nt;
  if (rc == BA_OK && zp) {
    ba_set_error("dense LDL': exactly zero pivot");
    return BA_ERR_ZERO_PIVOT;
  }
  return rc;
}
