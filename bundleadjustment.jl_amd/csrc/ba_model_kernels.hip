// Per-observation model kernels: reprojection residual, Jacobian pattern, hand-derived Jacobian values.
// gfx950 only.  One lane per observation; observations in BAL order (grouped by point), so the 3-double
// point block is shared by neighbouring lanes (one L1 line) and the 9-double camera block comes from the
// 72 B x ncams camera table, which lives in L2 (128 KB for Venice-1778).
//
// Replaces (reference paths): src/BALNLPModels.jl:11-36,39-55,115-122 (residual), :125-158 (pattern),
// :161-206 + src/JacobianByHand.jl:5-101 (values).  Compiled with -ffp-contract=off: Julia does not
// contract a*b+c, and the parity tests bound the distance to the reference evaluation order in ulps.
#include <cstdlib>

#include "ba_internal.h"

namespace {

constexpr int BLK = 256;

template <typename T>
struct Trig;
template <>
struct Trig<double> {
  static __device__ inline void sc(double th, double &s, double &c) { sincos(th, &s, &c); }
  static __device__ inline double sq(double v) { return sqrt(v); }
};
template <>
struct Trig<float> {
  // Julia's Float32 sin/cos round a double-precision kernel once; do the same.
  static __device__ inline void sc(float th, float &s, float &c) {
    double sd, cd;
    sincos((double)th, &sd, &cd);
    s = (float)sd;
    c = (float)cd;
  }
  static __device__ inline float sq(float v) { return sqrtf(v); }
};

// projection!  src/BALNLPModels.jl:17-33 in the reference's evaluation order (left folds).
// P1 is returned too (the Jacobian needs it).  No theta->0 / z==0 guards: the reference has none.
template <typename T>
__device__ inline void project(const T X[3], const T C[9], T P1[3], T P2[2], T out[2], T &th, T &s, T &c,
                               T k[3], T &d) {
  th = Trig<T>::sq(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
  k[0] = C[0] / th;
  k[1] = C[1] / th;
  k[2] = C[2] / th;
  Trig<T>::sc(th, s, c);
  T kx0 = k[1] * X[2] - k[2] * X[1];
  T kx1 = k[2] * X[0] - k[0] * X[2];
  T kx2 = k[0] * X[1] - k[1] * X[0];
  d = k[0] * X[0] + k[1] * X[1] + k[2] * X[2];
  T omc_d = (1 - c) * d;
  P1[0] = ((c * X[0] + s * kx0) + omc_d * k[0]) + C[3];
  P1[1] = ((c * X[1] + s * kx1) + omc_d * k[1]) + C[4];
  P1[2] = ((c * X[2] + s * kx2) + omc_d * k[2]) + C[5];
  P2[0] = -P1[0] / P1[2];
  P2[1] = -P1[1] / P1[2];
  T sqn = P2[0] * P2[0] + P2[1] * P2[1];
  // scaling_factor: the literal 1.0 is a Float64 (BALNLPModels.jl:13) -> promoted for T = Float32
  double sc = (1.0 + (double)(C[6] * sqn)) + (double)(C[7] * (sqn * sqn));
  double fs = (double)C[8] * sc;
  out[0] = (T)(fs * (double)P2[0]);
  out[1] = (T)(fs * (double)P2[1]);
}

template <typename T>
__global__ __launch_bounds__(BLK) void k_residual(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                   const int *__restrict__ pnt0, const T *__restrict__ x,
                                                   const T *__restrict__ pt2d, T *__restrict__ r) {
  int64_t o = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (o >= nobs) return;
  const T *Xp = x + 3 * (int64_t)pnt0[o];
  const T *Cp = x + 3 * npnts + 9 * (int64_t)cam0[o];
  T X[3], C[9];
#pragma unroll
  for (int i = 0; i < 3; i++) X[i] = Xp[i];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = Cp[i];
  T P1[3], P2[2], out[2], th, s, c, k[3], d;
  project<T>(X, C, P1, P2, out, th, s, c, k, d);
  // cx .-= pt2d  (BALNLPModels.jl:118)
  r[2 * o] = out[0] - pt2d[2 * o];
  r[2 * o + 1] = out[1] - pt2d[2 * o + 1];
}

// jac_structure!  BALNLPModels.jl:125-158.  One lane writes two consecutive entries (16 B) of rows and of cols.
__global__ __launch_bounds__(BLK) void k_jac_structure(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                        const int *__restrict__ pnt0, int64_t *__restrict__ rows,
                                                        int64_t *__restrict__ cols) {
  int64_t e = 2 * ((int64_t)blockIdx.x * BLK + threadIdx.x);
  if (e >= 24 * nobs) return;
  int64_t o = e / 24;
  int j = (int)(e - 24 * o);
  int jj = j >= 12 ? j - 12 : j;
  int64_t row = 2 * o + 1 + (j >= 12 ? 1 : 0);
  int64_t cp = 3 * (int64_t)pnt0[o] + 1;
  int64_t cc = 3 * npnts + 9 * (int64_t)cam0[o] + 1 - 3;
  longlong2 rv, cv;
  rv.x = row;
  rv.y = row;
  cv.x = (jj < 3) ? cp + jj : cc + jj;
  cv.y = (jj + 1 < 3) ? cp + jj + 1 : cc + jj + 1;
  *reinterpret_cast<longlong2 *>(rows + e) = rv;
  *reinterpret_cast<longlong2 *>(cols + e) = cv;
}

// 2x12 block of one observation: denseJ = (JP3*JP2)*JP1 evaluated without the structural zeros of the
// reference's padded 2x5 / 5x6 / 6x12 matrices (BALNLPModels.jl:177-197).  Entries of JP1/JP2/JP3 are
// computed in T, the chain products in double (for T = Float32 the reference's scratch matrices are
// Float64, BALNLPModels.jl:177,179).  Column order [X(3), r(3), t(3), k1, k2, f].
// Unlike the residual (which keeps the reference's operation order), this block is tolerance-matched by
// contract (the reference's own values depend on OpenBLAS' summation order, SURVEY.md 8c: <= 1e-12 of the
// block inf-norm), so the ten divisions by theta and z are two reciprocals and a*b+c contracts to FMA.
template <typename T>
__device__ inline void jac_block(const T X[3], const T C[9], T J[24]) {
#pragma clang fp contract(fast)
  const T x = X[0], y = X[1], z = X[2];
  const T th = Trig<T>::sq(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
  const T ith = (T)1 / th;
  const T kx = C[0] * ith, ky = C[1] * ith, kz = C[2] * ith;
  T s, c;
  Trig<T>::sc(th, s, c);
  const T d = kx * x + ky * y + kz * z;
  const T omc = 1 - c;
  const T omc_d = omc * d;
  T P1[3], P2[2];
  P1[0] = ((c * x + s * (ky * z - kz * y)) + omc_d * kx) + C[3];
  P1[1] = ((c * y + s * (kz * x - kx * z)) + omc_d * ky) + C[4];
  P1[2] = ((c * z + s * (kx * y - ky * x)) + omc_d * kz) + C[5];
  const T iz = (T)1 / P1[2];
  P2[0] = -P1[0] * iz;
  P2[1] = -P1[1] * iz;
  const T sth = s * ith, omcth = omc * ith;
  const T kx2 = kx * kx, ky2 = ky * ky, kz2 = kz * kz;
  // JP1!  JacobianByHand.jl:27-59: R = d(P1)/dX (3x3), G = d(P1)/dr (3x3)
  T R[3][3], G[3][3];
  R[0][0] = c + omc * kx2;
  R[0][1] = -s * kz + omc * ky * kx;
  R[0][2] = s * ky + omc * kz * kx;
  G[0][0] = -s * x * kx + c * kx * (ky * z - kz * y) + sth * (-ky * kx * z + kz * kx * y) + s * kx2 * d +
            omcth * (2 * x * kx * (1 - kx2) + y * ky * (1 - 2 * kx2) + z * kz * (1 - 2 * kx2));
  G[0][1] = -s * x * ky + c * ky * (ky * z - kz * y) + sth * ((1 - ky2) * z + kz * ky * y) + s * kx * ky * d +
            omcth * (-2 * x * kx2 * ky + y * kx * (1 - 2 * ky2) - 2 * z * kx * ky * kz);
  G[0][2] = -s * x * kz + c * kz * (ky * z - kz * y) + sth * (-ky * kz * z - (1 - kz2) * y) + s * kx * kz * d +
            omcth * (-2 * x * kx2 * kz - 2 * y * kx * ky * kz + z * kx * (1 - 2 * kz2));
  R[1][0] = s * kz + omc * ky * kx;
  R[1][1] = c + omc * ky2;
  R[1][2] = -s * kx + omc * ky * kz;
  G[1][0] = -s * y * kx + c * kx * (kz * x - kx * z) + sth * (-kz * kx * x - (1 - kx2) * z) + s * kx * ky * d +
            omcth * (x * ky * (1 - 2 * kx2) - 2 * y * kx * ky2 - 2 * z * kx * ky * kz);
  G[1][1] = -s * y * ky + c * ky * (kz * x - kx * z) + sth * (-kz * ky * x + kx * ky * z) + s * ky2 * d +
            omcth * (x * kx * (1 - 2 * ky2) + 2 * y * ky * (1 - ky2) + z * kz * (1 - 2 * ky2));
  G[1][2] = -s * y * kz + c * kz * (kz * x - kx * z) + sth * ((1 - kz2) * x + kx * kz * z) + s * kz * ky * d +
            omcth * (-2 * x * kx * ky * kz - 2 * y * ky2 * kz + z * ky * (1 - 2 * kz2));
  R[2][0] = -s * ky + omc * kx * kz;
  R[2][1] = s * kx + omc * ky * kz;
  R[2][2] = c + omc * kz2;
  G[2][0] = -s * z * kx + c * kx * (kx * y - ky * x) + sth * ((1 - kx2) * y + kx * ky * x) + s * kx * kz * d +
            omcth * (x * kz * (1 - 2 * kx2) - 2 * y * kx * ky * kz - 2 * z * kx * kz2);
  G[2][1] = -s * z * ky + c * ky * (kx * y - ky * x) + sth * (-kx * ky * y - (1 - ky2) * x) + s * ky * kz * d +
            omcth * (-2 * x * kx * ky * kz + y * kz * (1 - 2 * ky2) - 2 * z * ky * kz2);
  G[2][2] = -s * z * kz + c * kz * (kx * y - ky * x) + sth * (-kx * kz * y + kz * ky * x) + s * kz2 * d +
            omcth * (x * kx * (1 - 2 * kz2) + y * ky * (1 - 2 * kz2) + 2 * z * kz * (1 - kz2));
  // JP2!  JacobianByHand.jl:62-77
  const T a = -iz;
  const T b0 = P1[0] * iz * iz;
  const T b1 = P1[1] * iz * iz;
  // JP3!  JacobianByHand.jl:80-101
  const T k1 = C[6], k2 = C[7], f = C[8];
  const T xx = P2[0], yy = P2[1];
  const T norm2 = xx * xx + yy * yy;
  const T norm4 = norm2 * norm2;
  const T rr = 1 + k1 * norm2 + k2 * norm4;
  const T gx = 2 * k1 * xx + k2 * (4 * (xx * xx * xx) + 4 * xx * (yy * yy));
  const T gy = 2 * k1 * yy + k2 * (4 * (yy * yy * yy) + 4 * yy * (xx * xx));
  T JP3[2][5];
  JP3[0][0] = f * rr + f * gx * xx;
  JP3[0][1] = f * gy * xx;
  JP3[0][2] = f * norm2 * xx;
  JP3[0][3] = f * norm4 * xx;
  JP3[0][4] = rr * xx;
  JP3[1][0] = f * gx * yy;
  JP3[1][1] = f * rr + f * gy * yy;
  JP3[1][2] = f * norm2 * yy;
  JP3[1][3] = f * norm4 * yy;
  JP3[1][4] = rr * yy;
  const bool z_zero = (P1[2] == 0);  // P2()/JP2!() return NaN there: the whole block is NaN -> 0 (BALNLPModels.jl:201)
#pragma unroll
  for (int q = 0; q < 2; q++) {
    double M0 = (double)JP3[q][0] * (double)a;
    double M1 = (double)JP3[q][1] * (double)a;
    double M2 = (double)JP3[q][0] * (double)b0 + (double)JP3[q][1] * (double)b1;
    double v[12];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      v[cc] = (M0 * (double)R[0][cc] + M1 * (double)R[1][cc]) + M2 * (double)R[2][cc];
      v[3 + cc] = (M0 * (double)G[0][cc] + M1 * (double)G[1][cc]) + M2 * (double)G[2][cc];
    }
    v[6] = M0;
    v[7] = M1;
    v[8] = M2;
    v[9] = (double)JP3[q][2];
    v[10] = (double)JP3[q][3];
    v[11] = (double)JP3[q][4];
#pragma unroll
    for (int cc = 0; cc < 12; cc++) {
      T t = (T)v[cc];
      J[12 * q + cc] = (z_zero || t != t) ? (T)0 : t;
    }
  }
}

// jac_coord!: one lane per observation computes its 24 values in registers; each wave transposes its 64 x 24 values
// through its own LDS slot (row stride 25 elements: conflict-free ds_write_b64 / ds_read_b64) so that every wave store
// instruction writes 1 KiB of consecutive addresses.
// Each wave handles NBW = 4 consecutive batches of 64 observations, straight-line (no loop: inside a loop the compiler
// hoists the ~100 fp64 constants of sincos into registers, 240 VGPRs): the index pairs and the point/camera blocks of
// ALL its batches are requested up front, so the index -> gather latency is paid once per 256 observations and the
// stores of one batch drain while the next is computed.  Measured on Venice (5.0 M observations, 1.03 GB of HBM traffic
// by the TCC counters = the algorithmic bytes): NBW 1 / 2 / 4 -> 0.295 / 0.285 / 0.260 ms; the arithmetic is free
// (0.26 ms with it removed), the stores alone take 0.18 ms, the loads alone 0.07 ms.
// DBG (tools/bench_jac.py only): 1 = skip the arithmetic, 2 = skip the stores.
#ifndef BA_NBW
#define BA_NBW 4
#endif
constexpr int NBW = BA_NBW;

constexpr int CPAD = 16;  // camera rows are re-laid out to 16 elements (one 128-byte line for double) before the launch

// x's camera block (9 per camera, 8-byte aligned rows) -> padded, 16-byte aligned rows.  Why: a lane fetching its camera
// with nine 8-byte loads makes every one of the nine wave instructions pull 64 different cache lines through the CU's
// 64 B/clk L1 fill path, which the 12 KB of stores per batch need as well (measured: gathers and stores add up, 0.29 ms);
// five 16-byte loads of one aligned line fetch each line once.
template <typename T>
__global__ __launch_bounds__(BLK) void k_pad_cams(int64_t ncams, const T *__restrict__ cams, T *__restrict__ padded) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= ncams * CPAD) return;
  int64_t c = i / CPAD;
  int j = (int)(i - c * CPAD);
  padded[i] = j < 9 ? cams[9 * c + j] : (T)0;
}
template <typename T, int DBG = 0>
__global__ __launch_bounds__(BLK) void k_jac_coord(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                    const int *__restrict__ pnt0, const T *__restrict__ x,
                                                    const T *__restrict__ cpad, T *__restrict__ vals) {
  __shared__ T tile[BLK / 64][64 * 25];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w0 = ((int64_t)blockIdx.x * (BLK / 64) + wv) * (64 * NBW);  // first observation of this wave
  if (w0 >= nobs) return;
  constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte store
  T *ws = tile[wv];
  T X[NBW][3], C[NBW][9];
  int pi[NBW], ci[NBW];
#pragma unroll
  for (int q = 0; q < NBW; q++) {  // all index pairs first (one latency), then all gathers (one more)
    int64_t o = w0 + 64 * q + lane;
    o = o < nobs ? o : nobs - 1;
    pi[q] = pnt0[o];
    ci[q] = cam0[o];
  }
#pragma unroll
  for (int q = 0; q < NBW; q++) {
    const int p0 = pi[q], c0 = ci[q];
#pragma unroll
    for (int i = 0; i < 3; i++) X[q][i] = x[3 * (int64_t)p0 + i];
    {
      constexpr int VL = 16 / sizeof(T);  // elements per 16-byte load
      typedef T vt __attribute__((ext_vector_type(VL)));
      const vt *row = reinterpret_cast<const vt *>(cpad + CPAD * (int64_t)c0);
      T tmp[((9 + VL - 1) / VL) * VL];
#pragma unroll
      for (int u = 0; u < (9 + VL - 1) / VL; u++) {
        vt v = row[u];
#pragma unroll
        for (int e2 = 0; e2 < VL; e2++) tmp[u * VL + e2] = v[e2];
      }
#pragma unroll
      for (int i = 0; i < 9; i++) C[q][i] = tmp[i];
    }
  }
#pragma unroll
  for (int q = 0; q < NBW; q++) {
    const int64_t b0 = w0 + 64 * q;
    if (b0 >= nobs) break;
    T J[24];
    if (DBG & 1) {
#pragma unroll
      for (int j = 0; j < 24; j++) J[j] = X[q][j % 3] + C[q][j % 9];
    } else {
      jac_block<T>(X[q], C[q], J);
    }
    if (b0 + lane < nobs) {
#pragma unroll
      for (int j = 0; j < 24; j++) ws[lane * 25 + j] = J[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t nvalid = (nobs - b0) < 64 ? (nobs - b0) * 24 : 64 * 24;
    T *out = vals + b0 * 24;
#pragma unroll
    for (int it = 0; it < 24 / VEC; it++) {
      const int e = (it * 64 + lane) * VEC;
      if (e < nvalid && !((DBG & 2) && ws[0] != (T)12345.678)) {
        T v[VEC];
#pragma unroll
        for (int r = 0; r < VEC; r++) {
          const int ee = e + r, oo = ee / 24;
          v[r] = ws[oo * 25 + (ee - oo * 24)];
        }
        typedef T vst __attribute__((ext_vector_type(VEC)));
        vst w;
#pragma unroll
        for (int r = 0; r < VEC; r++) w[r] = v[r];
        // plain 16-byte store.  (An inline-asm `global_store_dwordx4 ... sc1` was tried to keep the stream out of L2: no
        // speed-up, and without the wait state hipcc inserts after a >64-bit store whose data registers are rewritten it
        // corrupted 0.5 % of the blocks at 12 M observations -- found by tools/dbg_big.py, hence the plain store.)
        *reinterpret_cast<vst *>(out + e) = w;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

static inline unsigned grid_for(int64_t n, int blk) { return (unsigned)((n + blk - 1) / blk); }
static inline unsigned jac_grid(int64_t nobs) { return grid_for(nobs, BLK * NBW); }

int launch_residual_f64(ba_problem *p, const double *d_x, double *d_r, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_RESIDUAL, st);
  hipLaunchKernelGGL(k_residual<double>, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_x, p->pt2d, d_r);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_residual_f32(ba_problem *p, const float *d_x, float *d_r, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_RESIDUAL, st);
  hipLaunchKernelGGL(k_residual<float>, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_x, p->pt2d_f32, d_r);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_jac_structure(ba_problem *p, int64_t *d_rows, int64_t *d_cols, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_JAC_STRUCTURE, st);
  hipLaunchKernelGGL(k_jac_structure, dim3(grid_for(12 * p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_rows, d_cols);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_jac_coord_f64(ba_problem *p, const double *d_x, double *d_vals, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_JAC_COORD, st);
  double *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  hipLaunchKernelGGL(k_pad_cams<double>, dim3(grid_for(p->ncams * CPAD, BLK)), dim3(BLK), 0, st, p->ncams,
                     d_x + 3 * p->npnts, cpad);
  hipLaunchKernelGGL(k_jac_coord<double>, dim3(jac_grid(p->nobs)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_x, (const double *)cpad, d_vals);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

extern "C" int ba_debug_jac_bench(ba_problem *p, const double *d_x, double *d_vals, int variant, int reps, double *ms_out) {
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipSetDevice(p->device));
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  hipStream_t st = p->stream;
  double *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  auto launch = [&]() {
    dim3 g(jac_grid(p->nobs)), b(BLK);
    hipLaunchKernelGGL(k_pad_cams<double>, dim3(grid_for(p->ncams * CPAD, BLK)), dim3(BLK), 0, st, p->ncams,
                       d_x + 3 * p->npnts, cpad);
#define JV(v) case v: hipLaunchKernelGGL((k_jac_coord<double, v>), g, b, 0, st, p->nobs, p->npnts, p->cam0, p->pnt0, d_x, (const double *)cpad, d_vals); break;
    switch (variant) {
      JV(1) JV(2) JV(3)
      default: hipLaunchKernelGGL((k_jac_coord<double, 0>), g, b, 0, st, p->nobs, p->npnts, p->cam0, p->pnt0, d_x, (const double *)cpad, d_vals);
    }
#undef JV
  };
  launch();
  BA_HIP_CHECK(hipStreamSynchronize(st));
  BA_HIP_CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; r++) launch();
  BA_HIP_CHECK(hipEventRecord(e1, st));
  BA_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / reps;
  return BA_OK;
}

int launch_jac_coord_f32(ba_problem *p, const float *d_x, float *d_vals, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_JAC_COORD, st);
  float *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  hipLaunchKernelGGL(k_pad_cams<float>, dim3(grid_for(p->ncams * CPAD, BLK)), dim3(BLK), 0, st, p->ncams,
                     d_x + 3 * p->npnts, cpad);
  hipLaunchKernelGGL(k_jac_coord<float>, dim3(jac_grid(p->nobs)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_x, (const float *)cpad, d_vals);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
