// Per-observation model kernels: reprojection residual, Jacobian pattern, hand-derived Jacobian values.
// gfx950 only.  One lane per observation; observations in BAL order (grouped by point), so the 3-double
// point block is shared by neighbouring lanes (one L1 line) and the 9-double camera block comes from the
// 72 B x ncams camera table, which lives in L2 (128 KB for Venice-1778).
//
// Replaces (reference paths): src/BALNLPModels.jl:11-36,39-55,115-122 (residual), :125-158 (pattern),
// :161-206 + src/JacobianByHand.jl:5-101 (values).  Compiled with -ffp-contract=off: Julia does not
// contract a*b+c, and the parity tests bound the distance to the reference evaluation order in ulps.
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
template <typename T>
__device__ inline void jac_block(const T X[3], const T C[9], T J[24]) {
  T P1[3], P2[2], out[2], th, s, c, kv[3], d;
  project<T>(X, C, P1, P2, out, th, s, c, kv, d);
  const T kx = kv[0], ky = kv[1], kz = kv[2];
  const T x = X[0], y = X[1], z = X[2];
  const T sth = s / th, omc = 1 - c, omcth = (1 - c) / th;
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
  const T a = -1 / P1[2];
  const T b0 = P1[0] / (P1[2] * P1[2]);
  const T b1 = P1[1] / (P1[2] * P1[2]);
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

// jac_coord!: one lane per observation computes its 24 values in registers; the block's 256 x 24 values are
// transposed through LDS (row stride 25 elements: conflict-free ds_write_b64 / ds_read_b64) so that every
// wave store instruction writes 1 KiB of consecutive addresses.
template <typename T>
__global__ __launch_bounds__(BLK) void k_jac_coord(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                    const int *__restrict__ pnt0, const T *__restrict__ x,
                                                    T *__restrict__ vals) {
  __shared__ T tile[BLK * 25];
  const int t = threadIdx.x;
  const int64_t o0 = (int64_t)blockIdx.x * BLK;
  const int64_t o = o0 + t;
  if (o < nobs) {
    const T *Xp = x + 3 * (int64_t)pnt0[o];
    const T *Cp = x + 3 * npnts + 9 * (int64_t)cam0[o];
    T X[3], C[9], J[24];
#pragma unroll
    for (int i = 0; i < 3; i++) X[i] = Xp[i];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = Cp[i];
    jac_block<T>(X, C, J);
#pragma unroll
    for (int j = 0; j < 24; j++) tile[t * 25 + j] = J[j];
  }
  __syncthreads();
  const int64_t nvalid = ((nobs - o0) < BLK ? (nobs - o0) : BLK) * 24;
  T *out = vals + o0 * 24;
  constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte store
#pragma unroll
  for (int it = 0; it < 24 / VEC; it++) {
    int e = (it * BLK + t) * VEC;
    if (e < nvalid) {
      T v[VEC];
#pragma unroll
      for (int q = 0; q < VEC; q++) {
        int ee = e + q;
        int oo = ee / 24;
        v[q] = tile[oo * 25 + (ee - oo * 24)];
      }
      if constexpr (sizeof(T) == 8) {
        double2 w;
        w.x = v[0];
        w.y = v[1];
        *reinterpret_cast<double2 *>(out + e) = w;
      } else {
        float4 w;
        w.x = v[0];
        w.y = v[1];
        w.z = v[2];
        w.w = v[3];
        *reinterpret_cast<float4 *>(out + e) = w;
      }
    }
  }
}

}  // namespace

static inline unsigned grid_for(int64_t n, int blk) { return (unsigned)((n + blk - 1) / blk); }

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
  hipLaunchKernelGGL(k_jac_coord<double>, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_x, d_vals);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_jac_coord_f32(ba_problem *p, const float *d_x, float *d_vals, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_JAC_COORD, st);
  hipLaunchKernelGGL(k_jac_coord<float>, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0,
                     p->pnt0, d_x, d_vals);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
