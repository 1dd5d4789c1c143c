// Per-observation model kernels: reprojection residual, Jacobian pattern, hand-derived Jacobian values.
// gfx950 only.  One lane per observation; observations in BAL order (grouped by point), so the 3-double
// point block is shared by neighbouring lanes (one L1 line) and the 9-double camera block comes from the
// 72 B x ncams camera table, which lives in L2 (128 KB for Venice-1778).
//
// Replaces (reference paths): src/BALNLPModels.jl:11-36,39-55,115-122 (residual), :125-158 (pattern),
// :161-206 + src/JacobianByHand.jl:5-101 (values).  Compiled with -ffp-contract=off: Julia does not
// contract a*b+c, and the parity tests bound the distance to the reference evaluation order in ulps.
#include <cstdlib>
#include <mutex>

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
// The part of the block that depends on the camera only (theta, the unit axis, sin and cos): evaluated once per camera
// by k_cam_pre instead of once per observation.  Row layout (CPAD = 16 elements, one 128-byte line for double):
//   [kx ky kz | t(3) | k1 k2 f | sin cos 1/theta | 0 0 0 0]
template <typename T>
__device__ inline void cam_pre(const T C[9], T P[12]) {
  // theta, the axis, sin and cos in the reference's operation order (BALNLPModels.jl:19-22: sqrt of the left-folded sum,
  // three divisions, no contraction; no theta -> 0 guard: the reference has none): the residual takes them from this
  // row, and evaluating them once per camera instead of once per observation changes no bit.  1/theta is for the
  // Jacobian only.
  const T th = Trig<T>::sq(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
  T s, c;
  Trig<T>::sc(th, s, c);
  P[0] = C[0] / th;
  P[1] = C[1] / th;
  P[2] = C[2] / th;
#pragma unroll
  for (int i = 3; i < 9; i++) P[i] = C[i];
  P[9] = s;
  P[10] = c;
  P[11] = (T)1 / th;
}

// projection!  src/BALNLPModels.jl:17-33 from a cam_pre row, in the reference's evaluation order (left folds; no z == 0
// guard: the reference has none).  scaling_factor: the literal 1.0 is a Float64 (BALNLPModels.jl:13) -> promoted for
// T = Float32.
template <typename T>
__device__ inline void project_pre(const T X[3], const T P[12], T out[2]) {
  const T k0 = P[0], k1 = P[1], k2 = P[2], s = P[9], c = P[10];
  T kx0 = k1 * X[2] - k2 * X[1];
  T kx1 = k2 * X[0] - k0 * X[2];
  T kx2 = k0 * X[1] - k1 * X[0];
  T d = k0 * X[0] + k1 * X[1] + k2 * X[2];
  T omc_d = (1 - c) * d;
  T P1[3], P2[2];
  P1[0] = ((c * X[0] + s * kx0) + omc_d * k0) + P[3];
  P1[1] = ((c * X[1] + s * kx1) + omc_d * k1) + P[4];
  P1[2] = ((c * X[2] + s * kx2) + omc_d * k2) + P[5];
  P2[0] = -P1[0] / P1[2];
  P2[1] = -P1[1] / P1[2];
  T sqn = P2[0] * P2[0] + P2[1] * P2[1];
  double sc = (1.0 + (double)(P[6] * sqn)) + (double)(P[7] * (sqn * sqn));
  double fs = (double)P[8] * sc;
  out[0] = (T)(fs * (double)P2[0]);
  out[1] = (T)(fs * (double)P2[1]);
}

template <typename T>
__device__ inline void jac_block(const T X[3], const T P[12], T J[24]) {
#pragma clang fp contract(fast)
  const T x = X[0], y = X[1], z = X[2];
  const T kx = P[0], ky = P[1], kz = P[2];
  const T s = P[9], c = P[10], ith = P[11];
  const T *C = P;  // C[3..8] = t, k1, k2, f as in the camera block
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
//
// Software-pipelined persistent waves.  The kernel is 93 % stores (192 of 205 bytes per observation) and a CU's vector
// memory instructions are served in order: a load issued behind the 12 KB-per-batch stores of the CU's other waves
// waits for them to drain, and the index -> gather dependency pays that wait twice.  With one batch per wave (or
// NBW = 4 straight-line batches, all loads up front) the two phases simply add up -- measured on Venice
// (5.0 M observations): stores alone 0.18 ms, + loads 0.26 ms, the arithmetic being free either way.  So every wave
// loops over batches b, b + W, b + 2W, ... and at the top of iteration i requests the index pairs of batch i + 2 and
// the point / camera rows of batch i + 1 (whose indices arrived an iteration ago) before it computes and stores batch
// i: no load is waited for in the iteration that issued it.  The trigonometric part of the block is per camera
// (cam_pre, k_cam_pre) -- which also keeps the loop body free of the ~100 fp64 sincos constants the compiler would
// hoist into registers.
// DBG (tools/bench_jac.py only): 1 = skip the arithmetic, 2 = skip the stores, 4 = no loads, 8 = no LDS transpose.
constexpr int CPAD = 16;  // precomputed camera rows: 16 elements (one 128-byte line for double), 16-byte aligned

// (Round 3, tried and removed: forming the rows INSIDE the consuming kernel -- its lowest workgroups first, everybody else
// polling a counter before gathering rows -- to save this 5.6 us launch and its gap.  With the readers counting themselves out
// so that the last one could reset the counters: 9 770 atomic adds on one address serialise at ~60 ns each, 594 us for a 57 us
// kernel.  With a counter that only grows and a per-launch target argument: correct and hipGraph-safe only outside recorded
// sequences, and 2 us SLOWER than the two launches back to back (49.2 against 47.3 us per residual at steady state,
// tools/bench_res.py): the wait costs more than the gap it removes.)
// x's camera block (9 per camera) -> cam_pre rows.  Besides hoisting the per-camera arithmetic, the padded rows make
// the gather five or six aligned 16-byte loads of ONE cache line per lane instead of nine 8-byte loads straddling two.
template <typename T>
__global__ __launch_bounds__(BLK) void k_cam_pre(int64_t ncams, const T *__restrict__ cams, T *__restrict__ pre) {
  int64_t c = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (c >= ncams) return;
  T C[9], P[12];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = cams[9 * c + i];
  cam_pre<T>(C, P);
#pragma unroll
  for (int i = 0; i < CPAD; i++) pre[CPAD * c + i] = i < 12 ? P[i] : (T)0;
}

template <typename T, int DBG>
__device__ inline void jac_load_idx(int64_t b, int lane, int64_t nobs, const int *__restrict__ cam0,
                                    const int *__restrict__ pnt0, int &pi, int &ci) {
  int64_t o = b * 64 + lane;
  o = o < nobs ? o : nobs - 1;  // batches past the end re-read the last observation: always a valid address
  pi = (DBG & 4) ? (int)(o & 1023) : pnt0[o];
  ci = (DBG & 4) ? (int)(o & 511) : cam0[o];
}

// Camera rows of one batch, requested line by line: CPAD / VL lanes share one observation and fetch the 16-byte pieces of
// ITS row (a full, aligned cache line per observation and instruction quad), instead of every lane walking its own
// row with six loads that each touch 64 different lines.  Measured on top of the 0.18 ms store stream of Venice: the
// lane-per-row gather costs +0.08 ms, index and point loads together +0.02 ms -- the L1's line-at-a-time tag path, which
// the stores need too, is what the divergent gather occupies.  The pieces land in registers (stg), cross to the lane
// that owns the observation through the wave's LDS slot (row stride CSTR dwords: conflict-free ds_read_b128).
template <typename T>
struct CamStage {
  static constexpr int VL = 16 / sizeof(T);       // elements per 16-byte piece
  static constexpr int PIECES = 12 / VL;           // pieces that carry data (6 double / 3 float)
  static constexpr int LPO = CPAD / VL;            // lanes per observation = pieces per row (8 / 4)
  static constexpr int OPI = 64 / LPO;             // observations per load instruction (8 / 16)
  static constexpr int NI = LPO;                   // load instructions per batch
  static constexpr int CSTR = sizeof(T) == 8 ? 28 : 20;  // LDS row stride in dwords
  typedef T vt __attribute__((ext_vector_type(VL)));
};

// The row requests are inline assembly on purpose.  Written as ordinary loads they are sunk by instruction selection to
// their first use -- behind the batch's stores -- and vmcnt retires in issue order, so the wave would wait for its own
// 12 KB of stores before it may touch the rows (measured: no gain over the unpipelined kernel).  As opaque asm the
// requests stay where they are written, ahead of the stores; jac_wait_cam<N> is their s_waitcnt, N = the VMEM
// instructions issued after them that may stay in flight.  Between the two the destination registers hold nothing:
// the asm operands tie them to the wait, so no compiler-generated use can precede it (tools/check_jac_isa.py checks the
// generated code for a stray read).  Compiler-counted waits elsewhere stay safe: an untracked older request only makes
// a counted wait stricter.
template <typename T, int DBG>
__device__ inline void jac_issue_cam(int ci, int lane, const T *__restrict__ cpre, typename CamStage<T>::vt stg[CamStage<T>::NI]) {
  typedef CamStage<T> S;
  const int piece = lane % S::LPO, sub = lane / S::LPO;
#pragma unroll
  for (int j = 0; j < S::NI; j++) {
    const int c = __shfl(ci, j * S::OPI + sub, 64);
    if (DBG & 16) {
#pragma unroll
      for (int e = 0; e < S::VL; e++) stg[j][e] = (T)(c + e);
    } else {  // every lane loads: the pieces past the data are the row's zero padding
      const typename S::vt *src = reinterpret_cast<const typename S::vt *>(cpre + CPAD * (int64_t)c) + piece;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(stg[j]) : "v"(src) : "memory");
    }
  }
}

template <typename T, int NAFTER>
__device__ inline void jac_wait_cam(typename CamStage<T>::vt stg[CamStage<T>::NI]) {
  if constexpr (CamStage<T>::NI == 8)
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7])
                 : "n"(NAFTER)
                 : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]) : "n"(NAFTER) : "memory");
}

template <typename T>
__device__ inline void jac_land_cam(const typename CamStage<T>::vt stg[CamStage<T>::NI], int lane, T *ws, T P[12]) {
  typedef CamStage<T> S;
  const int piece = lane % S::LPO, sub = lane / S::LPO;
  unsigned *w32 = reinterpret_cast<unsigned *>(ws);
  if (piece < S::PIECES) {
#pragma unroll
    for (int j = 0; j < S::NI; j++)
      *reinterpret_cast<typename S::vt *>(w32 + (j * S::OPI + sub) * S::CSTR + 4 * piece) = stg[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int u = 0; u < S::PIECES; u++) {
    const typename S::vt v = *reinterpret_cast<const typename S::vt *>(w32 + lane * S::CSTR + 4 * u);
#pragma unroll
    for (int e = 0; e < S::VL; e++) P[u * S::VL + e] = v[e];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // the slot is rewritten by the block transpose
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int DBG>
__device__ inline void jac_gather_point(int pi, const T *__restrict__ x, T X[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++) X[i] = (DBG & (4 | 32)) ? (T)(pi + i) : x[3 * (int64_t)pi + i];
}

// Block of one batch: 24 values per lane -> LDS transpose -> 16-byte stores of consecutive addresses.  FULL: all 64
// observations exist, nothing is predicated -- the pipelined loop needs that: with stores inside branches the compiler
// cannot count them and falls back to s_waitcnt vmcnt(0) before the next use of a prefetched register, which makes every
// wave wait for its own stores and serialises the load and store phases again.
template <typename T, int DBG, bool FULL>
__device__ inline void jac_emit(const T X0[3], const T P0[12], T *ws, int lane, int64_t b0, int64_t nobs,
                                T *__restrict__ vals) {
  constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte store
  T J[24];
  if (DBG & 1) {
#pragma unroll
    for (int j = 0; j < 24; j++) J[j] = X0[j % 3] + P0[j % 12];
  } else {
    jac_block<T>(X0, P0, J);
  }
  if ((FULL || b0 + lane < nobs) && !(DBG & 8)) {
#pragma unroll
    for (int j = 0; j < 24; j++) ws[lane * 25 + j] = J[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int64_t nvalid = FULL ? 64 * 24 : (nobs - b0) * 24;
  T *out = vals + b0 * 24;
#pragma unroll
  for (int it = 0; it < 24 / VEC; it++) {
    const int e = (it * 64 + lane) * VEC;
    if ((FULL || e < nvalid) && !(DBG & 2)) {
      typedef T vst __attribute__((ext_vector_type(VEC)));
      vst w;
#pragma unroll
      for (int r = 0; r < VEC; r++) {
        const int ee = e + r, oo = ee / 24;
        w[r] = (DBG & 8) ? J[(it * VEC + r) % 24] : ws[oo * 25 + (ee - oo * 24)];
      }
      // Non-temporal 16-byte store: the 0.96 GB stream is not kept in L2, where it would evict the camera rows and the
      // index / point lines every batch re-reads.  Measured on Venice, same box: plain stores 0.255-0.261 ms, nt stores
      // 0.192-0.196 ms.  DBG & 64 (bench only): plain stores.  (Use the builtin, not hand-written asm stores: an
      // inline-asm `global_store_dwordx4 ... sc1` without the wait state hipcc inserts after a >64-bit store whose
      // data registers are rewritten corrupted 0.5 % of the blocks at 12 M observations -- tools/dbg_big.py.)
      if (DBG & 64) *reinterpret_cast<vst *>(out + e) = w;
      else __builtin_nontemporal_store(w, reinterpret_cast<vst *>(out + e));
    }
  }
  if (DBG & 2) {  // keep the values alive without storing them
    T acc = 0;
#pragma unroll
    for (int j = 0; j < 24; j++) acc += J[j];
    if (acc == (T)12345.678) vals[0] = acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // the slot is rewritten by the next batch's camera rows
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int DBG = 0>
__global__ __launch_bounds__(BLK) void k_jac_coord(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                    const int *__restrict__ pnt0, const T *__restrict__ x,
                                                    const T *__restrict__ cpre, T *__restrict__ vals) {
  typedef CamStage<T> S;
  constexpr int CDBG = (DBG & 4) ? (DBG | 16) : DBG;
  constexpr int NST = (DBG & 2) ? 0 : (int)(24 * sizeof(T) / 16);  // store instructions of a full batch (12 / 6)
  __shared__ __attribute__((aligned(16))) T tile[BLK / 64][64 * 25];
  static_assert(64 * S::CSTR * 4 <= 64 * 25 * sizeof(T), "camera staging must fit the wave's transpose slot");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nfull = nobs / 64;                      // batches with all 64 observations
  const int64_t nw = (int64_t)gridDim.x * (BLK / 64);  // waves in the grid = batch stride of one wave
  const int64_t wid = (int64_t)blockIdx.x * (BLK / 64) + wv;
  T *ws = tile[wv];
  typename S::vt stg[S::NI];
  T X0[3], P0[12];
  if (wid < nfull) {
    int64_t b = wid;
    int pi1, ci1;
    {  // first batch: loaded and landed before the loop, so that nothing is pending where the loop is entered (a load
       // still in flight there would make the compiler wait for the loop's own stores at the same program point)
      int pi0, ci0;
      jac_load_idx<T, DBG>(b, lane, nobs, cam0, pnt0, pi0, ci0);
      jac_load_idx<T, DBG>(b + nw, lane, nobs, cam0, pnt0, pi1, ci1);
      jac_gather_point<T, DBG>(pi0, x, X0);
      jac_issue_cam<T, CDBG>(ci0, lane, cpre, stg);
      jac_wait_cam<T, 0>(stg);
      jac_land_cam<T>(stg, lane, ws, P0);
      // make the compiler's own wait for these loads happen here: pending at the loop header they would be merged with
      // the back edge's state into an s_waitcnt vmcnt(0) that also waits for the loop's stores
      asm volatile("" ::"v"(pi1), "v"(ci1), "v"(X0[0]), "v"(X0[1]), "v"(X0[2]));
    }
    for (; b < nfull; b += nw) {  // every wave's trip count is bounded by nfull: no wave waits on another
      int pi2, ci2;
      T X1[3];
      jac_load_idx<T, DBG>(b + 2 * nw, lane, nobs, cam0, pnt0, pi2, ci2);  // indices two batches ahead
      jac_gather_point<T, DBG>(pi1, x, X1);                                // point and camera rows one batch ahead,
      jac_issue_cam<T, CDBG>(ci1, lane, cpre, stg);                        // requested BEFORE this batch's stores
      jac_emit<T, DBG, true>(X0, P0, ws, lane, b * 64, nobs, vals);
      jac_wait_cam<T, NST>(stg);  // the stores just issued stay in flight
      jac_land_cam<T>(stg, lane, ws, P0);
#pragma unroll
      for (int i = 0; i < 3; i++) X0[i] = X1[i];
      pi1 = pi2;
      ci1 = ci2;
    }
  }
  if ((nobs & 63) && wid == nfull % nw) {  // the ragged last batch, by the wave whose turn it would have been
    int pi0, ci0;
    jac_load_idx<T, DBG>(nfull, lane, nobs, cam0, pnt0, pi0, ci0);
    jac_gather_point<T, DBG>(pi0, x, X0);
    jac_issue_cam<T, CDBG>(ci0, lane, cpre, stg);
    jac_wait_cam<T, 0>(stg);
    jac_land_cam<T>(stg, lane, ws, P0);
    jac_emit<T, DBG, false>(X0, P0, ws, lane, nfull * 64, nobs, vals);
  }
}

// cons!: one lane per observation.  Everything that depends on the camera alone (theta, axis, sin, cos: a square root,
// three divisions and a sincos) comes from the cam_pre row of the camera, computed once per camera by k_cam_pre at this x,
// and the rows reach the lanes the way k_jac_coord's do: eight (four) lanes fetch one observation's 128-byte (64-byte)
// row as aligned 16-byte pieces -- a full line per observation and instruction -- and hand it over through the wave's LDS
// slot.  The first version (nine scattered 8-byte loads and the whole trigonometric chain per lane) ran at 31 % of the
// HBM roofline on Venice.
// RNB batches of 64 observations per wave, straight-line: all index, point and row requests of the wave are issued before
// the first batch is evaluated.  Venice (5.0 M observations), rocprofv3: 91.9 us for the first version, 56.6-60.0 us with
// one batch per wave, 54.8-57.4 us with two (shipped), 77 us with four (registers); what bounds it now is the L1's
// request path (per 64 observations: 8 row instructions of 1 KB each beside 7 others), not HBM.
template <typename T, int RNB>
__global__ __launch_bounds__(BLK) void k_residual(int64_t nobs, const int *__restrict__ cam0, const int *__restrict__ pnt0,
                                                   const T *__restrict__ x, const T *__restrict__ cpre,
                                                   const T *__restrict__ pt2d, T *__restrict__ r) {
  typedef CamStage<T> S;
  typedef T v2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) unsigned slot[BLK / 64][64 * S::CSTR];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t base = ((int64_t)blockIdx.x * (BLK / 64) + wv) * (64 * RNB) + lane;
  const int piece = lane % S::LPO, sub = lane / S::LPO;
  int ci[RNB], pi[RNB];
  v2 m[RNB];
  T X[RNB][3];
  typename S::vt stg[RNB][S::NI];
#pragma unroll
  for (int b = 0; b < RNB; b++) {
    const int64_t o = base + 64 * b;
    const int64_t oc = o < nobs ? o : nobs - 1;  // lanes past the end shadow the last observation (their rows are still fetched)
    ci[b] = cam0[oc];
    pi[b] = pnt0[oc];
    m[b] = *reinterpret_cast<const v2 *>(pt2d + 2 * oc);
  }
#pragma unroll
  for (int b = 0; b < RNB; b++) {
#pragma unroll
    for (int i = 0; i < 3; i++) X[b][i] = x[3 * (int64_t)pi[b] + i];
#pragma unroll
    for (int j = 0; j < S::NI; j++) {
      const int c = __shfl(ci[b], j * S::OPI + sub, 64);
      // (every lane loads, the pieces past the data too: under a per-lane condition the requests serialise -- 165 us)
      stg[b][j] = *(reinterpret_cast<const typename S::vt *>(cpre + CPAD * (int64_t)c) + piece);
    }
  }
#pragma unroll
  for (int b = 0; b < RNB; b++) {
    T P[12], out[2];
    jac_land_cam<T>(stg[b], lane, reinterpret_cast<T *>(slot[wv]), P);
    project_pre<T>(X[b], P, out);
    const int64_t o = base + 64 * b;
    if (o < nobs) {  // cx .-= pt2d  (BALNLPModels.jl:118)
      v2 res;
      res[0] = out[0] - m[b][0];
      res[1] = out[1] - m[b][1];
      *reinterpret_cast<v2 *>(r + 2 * o) = res;
    }
  }
}

}  // namespace

static inline unsigned grid_for(int64_t n, int blk) { return (unsigned)((n + blk - 1) / blk); }
// Persistent grid of k_jac_coord: as many workgroups as the device keeps resident (LDS: 3 per CU for double), never
// more than there are batches of 64 observations.
template <typename K>
static unsigned jac_grid(ba_problem *p, K kernel, int64_t nobs) {
  // queried once per kernel and device (the launch may be recorded into a hipGraph: no runtime queries there)
  struct Entry { const void *k; int dev, per_cu, ncu; };
  static Entry cache[32];
  static int ncache = 0;
  static std::mutex mu;  // handles may be driven from several host threads (one rank per thread in the loopback tests)
  std::lock_guard<std::mutex> lock(mu);
  int per_cu = 0, ncu = 0;
  for (int q = 0; q < ncache; q++)
    if (cache[q].k == (const void *)kernel && cache[q].dev == p->device) {
      per_cu = cache[q].per_cu;
      ncu = cache[q].ncu;
    }
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || ncu < 1) ncu = 256;
    if (const char *e = getenv("BA_JAC_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    if (ncache < 32) cache[ncache++] = Entry{(const void *)kernel, p->device, per_cu, ncu};
  }
  const int64_t need = (nobs + 64 * (BLK / 64) - 1) / (64 * (BLK / 64));
  const int64_t cap = (int64_t)per_cu * ncu;
  return (unsigned)(need < cap ? need : cap);
}

int launch_residual_f64(ba_problem *p, const double *d_x, double *d_r, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_RESIDUAL, st);
  double *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  hipLaunchKernelGGL(k_cam_pre<double>, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_x + 3 * p->npnts,
                     cpad);
  hipLaunchKernelGGL((k_residual<double, 2>), dim3(grid_for(p->nobs, BLK * 2)), dim3(BLK), 0, st, p->nobs, p->cam0, p->pnt0, d_x,
                     (const double *)cpad, p->pt2d, d_r);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_residual_f32(ba_problem *p, const float *d_x, float *d_r, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_RESIDUAL, st);
  float *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  hipLaunchKernelGGL(k_cam_pre<float>, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_x + 3 * p->npnts, cpad);
  hipLaunchKernelGGL((k_residual<float, 2>), dim3(grid_for(p->nobs, BLK * 2)), dim3(BLK), 0, st, p->nobs, p->cam0, p->pnt0, d_x,
                     (const float *)cpad, p->pt2d_f32, d_r);
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
  hipLaunchKernelGGL(k_cam_pre<double>, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_x + 3 * p->npnts,
                     cpad);
  hipLaunchKernelGGL((k_jac_coord<double, 0>), dim3(jac_grid(p, k_jac_coord<double, 0>, p->nobs)), dim3(BLK), 0, st,
                     p->nobs, p->npnts, p->cam0, p->pnt0, d_x, (const double *)cpad, d_vals);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_jac_coord_f32(ba_problem *p, const float *d_x, float *d_vals, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  ProfScope ps(p, PC_JAC_COORD, st);
  float *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  hipLaunchKernelGGL(k_cam_pre<float>, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_x + 3 * p->npnts, cpad);
  hipLaunchKernelGGL((k_jac_coord<float, 0>), dim3(jac_grid(p, k_jac_coord<float, 0>, p->nobs)), dim3(BLK), 0, st, p->nobs,
                     p->npnts, p->cam0, p->pnt0, d_x, (const float *)cpad, d_vals);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
