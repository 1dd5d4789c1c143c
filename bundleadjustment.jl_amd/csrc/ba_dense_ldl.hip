// Dense blocked LDL' (no pivoting) of the reduced camera system on the gfx950 f64 matrix cores.
//
// Replaces the camera tail of the reference's scalar up-looking sparse LDL' (src/ldl_aux.jl:122-201, which is
// where 64-97 % of every reference iteration goes) and its triangular solves (src/ldl_aux.jl:4-42).  Like the
// reference it is an LDL' without pivoting that only fails on an exactly zero pivot (SQDException,
// src/ldl_aux.jl:199 -> BA_ERR_ZERO_PIVOT); S is symmetric positive definite in exact arithmetic.
//
// Storage: lower block triangle of NB x NB (128) tiles, each tile contiguous row-major (ba_internal.h).
// Right-looking, panels (tile columns) taken in pairs; the kernels of round 1's schedule (dense_ldl_factor has the fused
// pair schedule of round 2, k_ldl_pairdiag / k_ldl_pairtrsm, and dense_ldl_factor_dist the distributed one):
//   k_ldl_diag     : one workgroup factors tile (k,k) in LDS (L_kk, D_k) and forms L_kk^-1 explicitly;
//   k_ldl_trsm_rs  : X_i = S_ik L_kk^-T (= L_ik D_k) as an MFMA product with L_kk^-1 (32 rows per workgroup), stores
//                    V_i = X_i and L_ik = X_i D_k^-1; the forward substitution of the right-hand side rides along;
//   k_ldl_col_rs   : the update of the next tile column, which the second panel of the pair needs;
//   k_ldl_update<1>: S_ij -= V0_i L_jk' + V1_i L_{j,k+1}' for everything right of the pair, two panels per pass (v_mfma_f64_16x16x4_f64):
//                    the n^3/3 flops.
// Solves: forward substitution fused into the panel solves, diagonal scaling folded into the backward sweep by tile rows.
#include "ba_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

// scalar-type traits: the f64 and f32 forms of the 16x16x4 MFMA share the A/B operand lane map
// (A[i = l & 15][k = l >> 4]) but not the C/D map (f64: row = (l >> 4) + 4 reg; f32: row = 4 (l >> 4) + reg).
template <typename T>
struct RT;
template <>
struct RT<double> {
  typedef double v4 __attribute__((ext_vector_type(4)));
  typedef double v2 __attribute__((ext_vector_type(2)));
  static __device__ inline v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ inline int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct RT<float> {
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef float v2 __attribute__((ext_vector_type(2)));
  static __device__ inline v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ inline int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};
#define BA_VT typedef typename RT<T>::v4 d4; typedef typename RT<T>::v2 d2;

#ifndef BA_LDL_NT_C
#define BA_LDL_NT_C 0
#endif
constexpr int NT_C = BA_LDL_NT_C;  // 1: bulk update reads / writes its C tiles non-temporally (measured neutral: 38.97 vs 39.0 ms at n = 16002)
constexpr int KC = 16;       // K chunk of the GEMM kernels staged through LDS
constexpr int LDK = KC + 2;  // row stride 36 dwords: 36i+2k hit distinct banks for the MFMA operand reads
constexpr int DBUF = 0;  // 1: two LDS chunk buffers + one barrier per chunk; 0: one buffer + two barriers (measured faster:
                         // 54.4 vs 52.3 TFLOP/s on the pair update, the barrier is not the limiter and 2x LDS costs occupancy slack)
constexpr size_t GEMM_LDS_ELEMS = (size_t)(DBUF ? 2 : 1) * 2 * NB * LDK;


// ---- diagonal tile ---------------------------------------------------------------------------------------------
// One workgroup factors the 128x128 tile in LDS as 8x8 blocks of 16x16:
//   for each block column jb: unblocked LDL' of the 16x16 diagonal block (one wave, registers), the rows below it by
//   forward substitution, one row per thread (kept unscaled: X = L D), trailing blocks C(I,J) -= X(I) (X(J) D^-1)' by
//   MFMA -- with a look-ahead of one block column;
// and BESIDE that (round 3; see diag_tile) the eight 16x16 unit-lower inverses and the full unit-lower inverse of the tile,
// right-looking, in the strict upper block triangle of the LDS image.  Every finished block of the inverse goes to global
// memory at once; the factored tile itself is not written back (nothing reads it).  k_ldl_diag runs eight waves (two
// phases per block column), the fused pair kernel k_ldl_pairdiag four (three phases).
constexpr int LDA2 = 130;  // row stride 260 dwords = 4 mod 64: conflict-free MFMA operand reads
constexpr int XDL = 18;
constexpr int L16S = 18;   // row stride of the 16x16 multiplier block: 16-byte aligned rows for the row solves' paired reads
#ifndef BA_DIAG_THREADS
#define BA_DIAG_THREADS 512
#endif
constexpr int DIAG_THREADS = BA_DIAG_THREADS;  // k_ldl_diag: eight waves, two per SIMD (see diag_tile)
constexpr size_t DIAG_LDS_ELEMS = (size_t)(NB * LDA2 + 8 * 16 * XDL + 2 * NB + 16 * L16S);

// broadcast lane `src` (a compile-time constant after unrolling) of v to the whole wave: v_readlane_b32 into SGPRs
template <typename T>
__device__ inline T bcast_lane(T v, int src);
template <>
__device__ inline double bcast_lane<double>(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <>
__device__ inline float bcast_lane<float>(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// 1 / d on the pivot chain: hardware reciprocal + Newton steps (to a last-bit error of the true quotient; the full IEEE
// division sequence is ~3x longer and sits 128 times on the tile's critical path).  0 -> inf, as division.
template <typename T>
__device__ inline T fast_recip(T d);
template <>
__device__ inline double fast_recip<double>(double d) {
  double x = __builtin_amdgcn_rcp(d);
  if (d != 0.0) {
    x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
    x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  }
  return x;
}
template <>
__device__ inline float fast_recip<float>(float d) {
  float x = __builtin_amdgcn_rcpf(d);
  if (d != 0.0f) x = __builtin_fmaf(__builtin_fmaf(-d, x, 1.0f), x, x);
  return x;
}

// bounded wait of a hoisted workgroup for `need` tiles of the trailing update running beside it (k_ldl_update raises
// *wait_ready once per finished tile, release at agent scope).  Returns false when the wait was abandoned: after ~0.5 s of
// polling it gives up and reports through the pivot flag (value 2), so the wave always reaches its exit.
__device__ inline bool hoisted_wait(const int *wait_ready, int need, int *flag) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    int seen = 0;
    const bool abandoned = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2;  // an earlier tile gave up
    for (unsigned n = 0; n < (1u << 17) && !abandoned; n++) {
      // relaxed poll: an acquire at agent scope invalidates caches on every iteration, which the trailing update running
      // on the same XCD pays for (n = 40000: 421-426 ms against 411 ms without the hoist); the one acquire that matters
      // is the fence after the loop
      seen = __hip_atomic_load(wait_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= need) break;
      __builtin_amdgcn_s_sleep(127);
    }
    ok = seen >= need;
    if (!ok) *flag = 2;
  }
  __syncthreads();
  if (!ok) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every thread: the tiles' lines are re-read from memory
  return true;
}

// factor one diagonal tile in the workgroup's LDS image `sm` (DIAG_LDS_ELEMS elements): see the section comment above
template <typename T, int NW>
__device__ __forceinline__ void diag_tile(T *__restrict__ Skk, T *__restrict__ Linv_k, T *__restrict__ D_k, int *__restrict__ flag,
                                 unsigned long long *__restrict__ stamps, T *sm) {
  BA_VT
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;  // diagnostic phase timers (stamps != null only)
#define STAMP(slot)                                        \
  if (stamps) {                                            \
    unsigned long long tnow = __builtin_amdgcn_s_memtime(); \
    tacc[slot] += tnow - tlast;                            \
    tlast = tnow;                                          \
  }
  if (stamps) tlast = __builtin_amdgcn_s_memtime();
  T *a = sm, *xd = sm + NB * LDA2, *dd = xd + 8 * 16 * XDL, *dinv = dd + NB, *l16 = dinv + NB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keeps the task loops' branches scalar
  const int fr = lane & 15, fk = lane >> 4;
  // tile -> LDS: 16-byte loads, 16 in flight per thread (unpredicated: loads under a per-thread condition serialise).
  // The strict upper triangle of the LDS image is never read before the inverse phase writes it.
  constexpr int NT = 64 * NW, LQ = NB * NB / 2 / NT / 2;  // threads; 16-byte loads per thread and round (one round of
                                                          // twice as many: no faster, 7.1 vs 6.5 kcycles)
#pragma unroll
  for (int b0 = 0; b0 < 2; b0++) {
    d2 v[LQ];
#pragma unroll
    for (int q = 0; q < LQ; q++) v[q] = *reinterpret_cast<const d2 *>(Skk + 2 * ((b0 * LQ + q) * NT + tid));
#pragma unroll
    for (int q = 0; q < LQ; q++) {
      const int idx = 2 * ((b0 * LQ + q) * NT + tid), i = idx >> 7, j = idx & (NB - 1);
      *reinterpret_cast<d2 *>(a + i * LDA2 + j) = v[q];
    }
  }
  for (int idx = tid; idx < 8 * 16 * XDL; idx += NT) xd[idx] = 0.0;
  __syncthreads();
  STAMP(0)
  // unblocked LDL' of the 16x16 diagonal block jb in the registers of one wave: lane i holds row i, a pivot and the
  // pivot column travel by v_readlane (constant lane numbers after unrolling) -- no LDS round trip and no barrier
  // inside the 16 dependent steps (the LDS version with a workgroup barrier per pivot took 3 us per block).
  auto pivots = [&](int jb) {
    const int o = 16 * jb;
    const int i = lane & 15;  // lanes 16..63 shadow lanes 0..15
    T r[16], invs[16];
#pragma unroll
    for (int c = 0; c < 16; c++) r[c] = a[(o + i) * LDA2 + o + c];
    T myd = 0, myinv = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const T d = bcast_lane<T>(r[j], j);
      const T inv = fast_recip<T>(d);
      invs[j] = inv;
      if (i == j) {
        myd = d;
        myinv = inv;
      }
      const T t = r[j] * inv;
#pragma unroll
      for (int c = j + 1; c < 16; c++) r[c] -= t * bcast_lane<T>(r[j], c);  // rows above the diagonal carry garbage, unused
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < i) a[(o + i) * LDA2 + o + c] = r[c];               // X = L D stays unscaled in the tile image
        l16[i * L16S + c] = (c < i) ? r[c] * invs[c] : (T)0;      // scaled copy for the row solves below (broadcast reads)
      }
      dd[o + i] = myd;
      dinv[o + i] = myinv;
      if (myd == (T)0) *flag = 1;
    }
  };
  // Float64 (round 3): the same 16 x 16 factorisation with the rank-1 updates of a block batched four at a time into ONE
  // v_mfma_f64_16x16x4_f64.  The wave holds the block in the matrix instruction's own accumulator layout -- lane
  // (i = l & 15, g = l >> 4) keeps the four entries A[i][g + 4 reg] of row i -- so a panel of four columns (4p .. 4p+3) IS
  // register p of the 64 lanes, already in the A / B operand layout [row = l & 15][k = l >> 4].  Per panel: the four panel
  // entries of every row are gathered into all four lanes of the row (one trip through the wave's LDS scratch), four scalar
  // steps touch the panel's own columns only (pivot and pivot column by v_readlane, as in the rank-1 form, but at most three
  // updates per step instead of fifteen), and one MFMA subtracts sum_k X[:,k] (X[:,k] / d_k)' from every column right of
  // the panel (A operand zeroed for the rows <= 4p+3: in the symmetric accumulator that leaves the columns already factored
  // untouched; rows above the diagonal carry garbage, unused, as before).  The sums differ from the rank-1 order in the last
  // bits only.  Measured (tools/bench_diag.py): pivot phases 18.9 -> 15.4 us, tile 51.3 -> 47.8 us; what is left per step is
  // the dependent chain itself -- v_readlane, v_rcp_f64, two Newton steps, multiply, one FMA: ~200 cycles at ~20 per dependent
  // Float64 operation (v_rcp_f64 alone is good to 4.6e-8, one Newton step to 20 ulp, two to the last bit:
  // tools/probes/rcp_f64.hip).  Two variants measured slower than the rank-1 form (51.3 us) and dropped: no gather, the two
  // cross-lane values of every step by ds_bpermute (55.8 us: four LDS crossbar trips per step sit on the chain); and the
  // sixteen reciprocals kept in an array indexed by the lane group (53.8 us: the compiler puts it in scratch memory).
  auto pivots_mfma = [&](int jb) {
    const int o = 16 * jb;
    const int i = lane & 15, g = lane >> 4;
    T *scr = l16;  // free until this block's multipliers are written at the end (row_solves(jb - 1) is behind a barrier)
    d4 R;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) R[reg] = a[(o + i) * LDA2 + o + g + 4 * reg];
    // inv_own[pp]: 1 / d of MY column of panel pp (column 4 pp + g), picked up as the steps go by
    T inv_own[4] = {0, 0, 0, 0}, myd = 0, myinv = 0;
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      scr[i * L16S + g] = R[pp];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      T P[4];
      {
        const d2 p01 = *reinterpret_cast<const d2 *>(scr + i * L16S), p23 = *reinterpret_cast<const d2 *>(scr + i * L16S + 2);
        P[0] = p01[0];
        P[1] = p01[1];
        P[2] = p23[0];
        P[3] = p23[1];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = 4 * pp + k;
        const T d = bcast_lane<T>(P[k], c);  // A[c][c]: row c's copy of the panel (lane c)
        const T inv = fast_recip<T>(d);
        if (g == k) inv_own[pp] = inv;
        if (i == c) {
          myd = d;
          myinv = inv;
        }
        const T t = P[k] * inv;
#pragma unroll
        for (int k2 = k + 1; k2 < 4; k2++) P[k2] -= t * bcast_lane<T>(P[k], 4 * pp + k2);
      }
      {
        T sel = P[0];
        if (g == 1) sel = P[1];
        if (g == 2) sel = P[2];
        if (g == 3) sel = P[3];
        R[pp] = sel;
      }
      if (pp < 3) {
        const T aop = (i > 4 * pp + 3) ? -R[pp] : (T)0;
        const T bop = R[pp] * inv_own[pp];
        R = RT<T>::mfma(aop, bop, R);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the scratch rows are rewritten by the next panel's gather
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane (i, g) holds X[i][c], c = g + 4 reg: the unscaled entries go back to the tile image, the scaled copy to l16
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int c = g + 4 * reg;
      if (c < i) a[(o + i) * LDA2 + o + c] = R[reg];
      l16[i * L16S + c] = (c < i) ? R[reg] * inv_own[reg] : (T)0;
    }
    if (g == 0) {
      dd[o + i] = myd;
      dinv[o + i] = myinv;
      if (myd == (T)0) *flag = 1;
    }
  };
  // X = A(:,jb) L16^-T for the rows below the diagonal block by forward substitution, one row per thread (the rows
  // are independent; X = L D stays unscaled).
  auto row_solves = [&](int jb) {
    const int o = 16 * jb;
    if (tid < NB - o - 16) {
      const int r = o + 16 + tid;
      // the 120 multipliers are the same for every row: all of them are requested up front (16-byte LDS reads, the
      // same address in every lane), then the substitution runs column by column out of registers -- element c still
      // receives its subtractions in the order m = 0, 1, ..., c-1 (3 340 -> 2 120 cycles per block column)
      T l[16][16];
#pragma unroll
      for (int c = 1; c < 16; c++)
#pragma unroll
        for (int m = 0; m < c; m += 2) {
          const d2 v = *reinterpret_cast<const d2 *>(l16 + c * L16S + m);
          l[c][m] = v[0];
          if (m + 1 < c) l[c][m + 1] = v[1];
        }
      T xr[16];
#pragma unroll
      for (int c = 0; c < 16; c++) xr[c] = a[r * LDA2 + o + c];
#pragma unroll
      for (int m = 0; m < 15; m++)
#pragma unroll
        for (int c = m + 1; c < 16; c++) xr[c] -= xr[m] * l[c][m];
#pragma unroll
      for (int c = 0; c < 16; c++) a[r * LDA2 + o + c] = xr[c];
    }
  };
  // C(I,J) -= X(I) * (X(J) D^-1)' with the X of block column jb
  auto block_update = [&](int jb, int I, int J) {
    const int o = 16 * jb;
    d4 acc;
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = a[(16 * I + RT<T>::row(lane, g)) * LDA2 + 16 * J + fr];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int k = o + 4 * kk + fk;
      acc = RT<T>::mfma(-a[(16 * I + fr) * LDA2 + k], a[(16 * J + fr) * LDA2 + k] * dinv[k], acc);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) a[(16 * I + RT<T>::row(lane, g)) * LDA2 + 16 * J + fr] = acc[g];
  };
  // Look-ahead of one block column: once block column jb+1 has received the update of block column jb, one wave factors
  // its diagonal block while the other three apply update jb to the block columns right of it; only
  // update(column jb+1) -> pivots(jb+1) -> row solves(jb+1) remains a dependent chain.  Every block still receives its
  // updates in the order jb = 0, 1, ...: the arithmetic is that of the plain right-looking loop, bit for bit
  // (tools/bench_diag.py prints a fingerprint of the outputs; 63.1 -> 58.1 us, with the row solves above 53.4 us).
  // (Tried and dropped: one accumulator per k-slice in block_update and in the inverse below -- four independent MFMA
  // chains instead of one dependent chain: slower, 60 us; the chain is not what these phases wait for.)
  // The inverse of the tile's unit-lower factor rides along with the factorisation (round 3; before, it was a phase of its
  // own after the last block column: 11.7 of the tile's 47.5 us, one wave's chain of dependent matrix instructions per
  // block column, followed by 4.5 us of stores).  X = L^-1, right-looking by block columns of L:
  //   X(J,J) = L16(J)^-1,   X(I,J) = -L16(I)^-1 acc(I,J),   acc(I,J) = sum_{K=J}^{I-1} L(I,K) X(K,J).
  // Stage s (block column s of L is final after row_solves(s), L16(s)^-1 comes from one wave beside row_solves(s), which
  // never occupies more than 112 threads):
  //   with the update of block column s+1 (7-s blocks): X(s,J) = -L16(s)^-1 acc(s,J), J < s (s blocks) -- seven tasks;
  //   while wave 0 factors diagonal block s+1: acc(I,J) += L(I,s) X(s,J), I > s, J <= s, dealt to the worker waves with
  //   the trailing blocks of the factorisation -- 28, 27, 25, 22, 18, 13, 7 independent tasks of four matrix
  //   instructions each, where the factorisation alone had 21, 15, 10, 6, 3, 1, 0.
  // acc(I,J) waits in the LDS slot X(I,J) will take (X(I,J)[i][j] at a[16J+j][16I+i]: the strict upper block triangle,
  // which the factorisation never touches), same lane, same register, and receives its terms in the order K = J, J+1,
  // ...: the sums are those of the former phase, bit for bit.  Every finished block goes to the Linv tile in global memory
  // at once: no store phase is left either.  The kernel k_ldl_diag runs this with eight waves (seven workers, two waves
  // per SIMD so that one's LDS round trips hide behind the other's matrix instructions).
  auto inv16 = [&](int jb) {  // lanes 0..15 of one wave, one column each: l[i][m] = a[i][m] * dinv[m]
    const int o = 16 * jb, c = lane;
    T *xj = xd + jb * 16 * XDL;
    // column-oriented: once x[m] is known every later row takes its term -- sixteen dependent steps of one multiply-add
    // each instead of a chain through all 120 (each row still sums its terms in the order m = 0, 1, ...: same bits)
    // (the column of step m + 1 and all sixteen 1/d are requested ahead of step m's arithmetic, and the scheduler is kept
    // from sinking those reads behind the value they would then wait for: without it every step pays an LDS round trip)
    T sacc[16], di[16], col[16], nxt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      sacc[i] = 0;
      di[i] = dinv[o + i];
      col[i] = (i >= 1) ? a[(o + i) * LDA2 + o] : (T)0;
      nxt[i] = 0;
    }
#pragma unroll
    for (int m = 0; m < 16; m++) {
#pragma unroll
      for (int i = m + 2; i < 16; i++) nxt[i] = a[(o + i) * LDA2 + o + m + 1];
      __builtin_amdgcn_sched_barrier(0);
      const T xv = (m < c) ? 0.0 : (m == c ? 1.0 : -sacc[m]);
      const T y = xv * di[m];  // dinv[m] * x[m]
#pragma unroll
      for (int i = m + 1; i < 16; i++) sacc[i] += col[i] * y;
      xj[m * XDL + c] = xv;
      Linv_k[(o + m) * NB + o + c] = xv;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = m + 2; i < 16; i++) col[i] = nxt[i];
    }
  };
  auto inv_finalize = [&](int I, int J) {  // X(I,J) = -L16(I)^-1 acc(I,J)
    d4 acc;
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = a[(16 * J + fr) * LDA2 + 16 * I + RT<T>::row(lane, g)];
    d4 out = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; g++) out = RT<T>::mfma(xd[(I * 16 + fr) * XDL + RT<T>::row(lane, g)], acc[g], out);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      a[(16 * J + fr) * LDA2 + 16 * I + RT<T>::row(lane, g)] = -out[g];
      Linv_k[(16 * I + RT<T>::row(lane, g)) * NB + 16 * J + fr] = -out[g];
    }
  };
  auto inv_update = [&](int I, int J, int K) {  // acc(I,J) += L(I,K) X(K,J)
    d4 acc = {0, 0, 0, 0};
    if (K > J) {
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = a[(16 * J + fr) * LDA2 + 16 * I + RT<T>::row(lane, g)];
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int k = 16 * K + 4 * kk + fk;
      const T af = a[(16 * I + fr) * LDA2 + k] * dinv[k];
      // X(K,J): the diagonal block's inverse lives in xd, the others transposed in the upper triangle (one address
      // computation for both: a branch here would split the four matrix instructions into four basic blocks)
      const T *bp = (K == J) ? xd + (J * 16 + fk) * XDL + fr : a + (16 * J + fr) * LDA2 + 16 * K + fk;
      const int bstep = (K == J) ? 4 * XDL : 4;
      acc = RT<T>::mfma(af, bp[kk * bstep], acc);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) a[(16 * J + fr) * LDA2 + 16 * I + RT<T>::row(lane, g)] = acc[g];
  };
  constexpr int NWK = NW - 1;  // worker waves 1 .. NW-1
  auto do_pivots = [&](int jb) {
    if constexpr (sizeof(T) == 8) pivots_mfma(jb);  // (the Float32 matrix instruction has another accumulator layout: rank-1 form)
    else pivots(jb);
  };
  if (wv == 0) do_pivots(0);
  __syncthreads();
  STAMP(1)
  row_solves(0);
  if (wv == 3 && lane < 16) inv16(0);
  __syncthreads();
  STAMP(2)
  if constexpr (NW == 8) {
    // Eight waves: two phases per block column, P(s) in the shadow of wave 0's chain (update of diagonal block s+1, then its
    // factorisation) and A(s+1) in the shadow of the row solves -- the update of block column s+1 no longer has a phase
    // and a barrier of its own.  Tasks (four matrix instructions each): F(I,J,s) trailing update of the factorisation with
    // block column s; X(s,J) = -L16(s)^-1 acc(s,J); U(I,J,K) acc(I,J) += L(I,K) X(K,J).
    //   P(s), seven workers:  X(s,J), J < s;  F(I,s+1,s), I > s+1 (the row solves of A(s+1) need them);
    //                         U(I,J,s-1), I > s, J < s;  F(I,J,s), J > s+1, as far as three rounds of tasks go;
    //   A(s+1), the five waves beside row solves and inv16:  U(s+1,J,s), J <= s (X(s+1,.) is due in P(s+1));  the rest of F.
    // Every block still receives its terms in the order K = 0, 1, ...: same bits as the four-wave form below.
    // (Tried and dropped: a worker's tasks two at a time -- two independent accumulators, loads issued together, the matrix
    // instructions of one chain in the other's latency: 36.0 us against 34.3, the workers' busy time went UP by ~10 %.  The
    // tasks are not waiting for their own latency: eight waves share one LDS pipe (~10 KB of operand traffic per task).)
    constexpr int CAP = 3 * NWK - 1;  // the wave that shares wave 0's SIMD sits out the third round (it was the last to
                                      // arrive by ~700 cycles per stage)
    // workers: the wave that shares wave 0's SIMD comes last; helpers of A: the two waves of the free SIMD first, the
    // one that shares inv16's SIMD last
    const int u = (wv == 4) ? 6 : (wv < 4 ? wv - 1 : wv - 2);
    const int h = (wv == 2) ? 0 : (wv == 6) ? 1 : (wv == 4) ? 2 : (wv == 5) ? 3 : (wv == 7) ? 4 : -1;
    auto f_task = [&](int s, int t) {  // t-th block of (I, J), s + 2 <= J <= I < 8
      int ii = 0;
      while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
      const int jj = t - ii * (ii + 1) / 2;
      block_update(s, s + 2 + ii, s + 2 + jj);
    };
    for (int s = 0; s < 7; s++) {
      const int nX = s, nC = 6 - s, nU = s * (7 - s), nF = (6 - s) * (7 - s) / 2;
      const int nfix = nX + nC + nU, nP = nfix + nF < CAP ? nfix + nF : (nfix > CAP ? nfix : CAP);
      unsigned long long c0 = 0;
      if (stamps) c0 = __builtin_amdgcn_s_memtime();
      if (wv == 0) {
        block_update(s, s + 1, s + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        do_pivots(s + 1);
      } else {
        for (int r = 0; r < 3 + (nfix + NWK - 1) / NWK; r++) {
          // rounds 0 and 1: seven tasks; from round 2 on: six
          const int t = r < 2 ? NWK * r + u : 2 * NWK + (NWK - 1) * (r - 2) + u;
          if ((r >= 2 && u == NWK - 1) || t >= nP) continue;
          if (t < nX) {
            inv_finalize(s, t);
          } else if (t < nX + nC) {
            block_update(s, s + 2 + (t - nX), s + 1);
          } else if (t < nfix) {
            const int e = t - nX - nC;
            inv_update(s + 1 + e % (7 - s), e / (7 - s), s - 1);
          } else {
            f_task(s, t - nfix);
          }
        }
      }
      if (stamps && lane == 0) stamps[6 + s * 8 + wv] = __builtin_amdgcn_s_memtime() - c0;  // busy time of this wave
      __syncthreads();
      STAMP(1)
      if (stamps) c0 = __builtin_amdgcn_s_memtime();
      row_solves(s + 1);
      if (wv == 3 && lane < 16) inv16(s + 1);
      if (h >= 0) {
        const int nA = (s + 1) + (nfix + nF - nP);
        for (int t = h; t < nA; t += 5) {
          if (t <= s) inv_update(s + 1, t, s);
          else f_task(s, nP - nfix + (t - (s + 1)));
        }
      }
      if (stamps && lane == 0) stamps[62 + s * 8 + wv] = __builtin_amdgcn_s_memtime() - c0;  // busy time in A(s+1)
      __syncthreads();
      STAMP(2)
    }
  } else {
    for (int jb = 0; jb < 7; jb++) {
      for (int t = wv; t < 7; t += NW) {  // seven tasks: block column jb + 1 of the factorisation, block row jb of the inverse
        if (t < 7 - jb) block_update(jb, jb + 1 + t, jb + 1);
        else inv_finalize(jb, t - (7 - jb));
      }
      __syncthreads();
      STAMP(3)
      unsigned long long c0 = 0;
      if (stamps) c0 = __builtin_amdgcn_s_memtime();
      if (wv == 0) {
        do_pivots(jb + 1);
      } else {
        // trailing blocks (I, J), jb + 2 <= J <= I < 8, of the factorisation, then the inverse's (I, J), I > jb >= J
        const int mb = 6 - jb, nblk = mb * (mb + 1) / 2, ninv = (7 - jb) * (jb + 1);
        for (int t = wv - 1; t < nblk + ninv; t += NWK) {
          if (t < nblk) {
            int ii = 0;
            while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
            const int jj = t - ii * (ii + 1) / 2;
            block_update(jb, jb + 2 + ii, jb + 2 + jj);
          } else {
            const int e = t - nblk, J = e / (7 - jb), I = jb + 1 + e % (7 - jb);
            inv_update(I, J, jb);
          }
        }
      }
      if (stamps && lane == 0) stamps[6 + jb * 8 + wv] = __builtin_amdgcn_s_memtime() - c0;  // busy time of this wave
      __syncthreads();
      STAMP(1)
      row_solves(jb + 1);
      if (wv == 3 && lane < 16) inv16(jb + 1);
      __syncthreads();
      STAMP(2)
    }
  }
  // last block row of the inverse: X(7,J) = -L16(7)^-1 acc(7,J)
  for (int J = wv; J < 7; J += NW) inv_finalize(7, J);
  if (tid < NB) D_k[tid] = dd[tid];
  STAMP(4)
  if (stamps && tid == 0)
    for (int q = 0; q < 6; q++) stamps[q] = tacc[q];
#undef STAMP
}

template <typename T>
__global__ __launch_bounds__(DIAG_THREADS) void k_ldl_diag(T *__restrict__ Skk, T *__restrict__ Linv_k,
                                                   T *__restrict__ D_k, int *__restrict__ flag,
                                                   unsigned long long *__restrict__ stamps,
                                                   const int *__restrict__ wait_ready, int clear_flag = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  // first kernel of an in-order factorisation: the pivot flag is cleared here instead of by a memset node of its own (the
  // first pivot is behind the tile's load and a barrier)
  if (clear_flag && threadIdx.x == 0) *flag = 0;
  // hoisted launch: started early (while CUs were free), waits in place until the trailing update running beside it has
  // finished this tile
  if (wait_ready && !hoisted_wait(wait_ready, 1, flag)) return;
  diag_tile<T, DIAG_THREADS / 64>(Skk, Linv_k, D_k, flag, stamps, reinterpret_cast<T *>(smraw));
}

// two diagonal tiles per launch (two independent runs of the block-sparse schedule: see "two runs per launch" below)
template <typename T>
__global__ __launch_bounds__(DIAG_THREADS) void k_ldl_diag2(T *__restrict__ SkkA, T *__restrict__ LinvA, T *__restrict__ DA,
                                                    T *__restrict__ SkkB, T *__restrict__ LinvB, T *__restrict__ DB,
                                                    int *__restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const bool second = blockIdx.x == 1;
  diag_tile<T, DIAG_THREADS / 64>(second ? SkkB : SkkA, second ? LinvB : LinvA, second ? DB : DA, flag, (unsigned long long *)nullptr,
                                  reinterpret_cast<T *>(smraw));
}

// ---- 128 x 128 x (128 NP) tile product C = sum_p A_p * B_p' on the matrix cores ------------------------------------
// A_p, B_p: contiguous row-major 128x128 tiles in global memory.  256 threads = 4 waves, wave w owns the 64x64
// quadrant (w >> 1, w & 1) as 4x4 MFMA blocks (128 accumulator VGPRs).  K is consumed in chunks of KC = 16 staged
// through two LDS buffers (row stride 18 doubles: conflict-free operand reads), one barrier per chunk: while chunk c is
// multiplied, chunk c+1 moves registers -> other buffer and the global loads of chunk c+2 are in flight; two
// workgroups per CU cover each other's barriers.
// YACC (forward substitution fused into the panel solve): threads 0..127 also accumulate yacc = sum_k B0[tid][k] bk[k]
// from the B chunks as they pass through LDS.
template <typename T, int NP, bool YACC = false>
__device__ inline void tile_gemm_abt(const T *__restrict__ A0, const T *__restrict__ B0,
                                     const T *__restrict__ A1, const T *__restrict__ B1, T *sA, T *sB,
                                     typename RT<T>::v4 acc[4][4], const T *__restrict__ bk = nullptr, T *yacc = nullptr) {
  BA_VT
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  const int fr = lane & 15, fk = lane >> 4;
  constexpr int NLD = (NB * KC / 2) / 256;  // 16-byte loads per thread per operand per chunk
  d2 pa[NLD], pb[NLD];
  constexpr int UPR = KC / 2;                 // 16-byte units per row of a chunk
  constexpr int RPS = 256 / UPR;              // rows covered by one step of the 256 threads
  const int lrow = tid / UPR, lc2 = tid % UPR;  // thread's first (row, 16-byte column) of a chunk
#pragma unroll
  for (int it = 0; it < NLD; it++) {
    pa[it] = *reinterpret_cast<const d2 *>(A0 + (lrow + RPS * it) * NB + 2 * lc2);
    pb[it] = *reinterpret_cast<const d2 *>(B0 + (lrow + RPS * it) * NB + 2 * lc2);
  }
  constexpr int NCH = NP * (NB / KC);
  constexpr int BUF = DBUF ? 2 * NB * LDK : 0;  // doubles per LDS buffer (A | B); sA/sB point at buffer 0
  if (DBUF) {  // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
#pragma unroll
    for (int it = 0; it < NLD; it++) {
      *reinterpret_cast<d2 *>(sA + (lrow + RPS * it) * LDK + 2 * lc2) = pa[it];
      *reinterpret_cast<d2 *>(sB + (lrow + RPS * it) * LDK + 2 * lc2) = pb[it];
    }
    if (NCH > 1) {
#pragma unroll
      for (int it = 0; it < NLD; it++) {
        pa[it] = *reinterpret_cast<const d2 *>(A0 + (lrow + RPS * it) * NB + KC + 2 * lc2);
        pb[it] = *reinterpret_cast<const d2 *>(B0 + (lrow + RPS * it) * NB + KC + 2 * lc2);
      }
    }
  }
  for (int ch = 0; ch < NCH; ch++) {
    // DBUF: buffer ch&1 is complete and everybody has finished reading the other one.  Single buffer: everybody has
    // finished reading the previous chunk.
    __syncthreads();
    T *cA = sA + (ch & 1) * BUF, *cB = sB + (ch & 1) * BUF;
    if (DBUF ? (ch + 1 < NCH) : true) {  // registers -> LDS (DBUF: chunk ch+1 into the other buffer, overlapping the MFMAs)
      T *nA = sA + ((ch + 1) & 1) * BUF, *nB = sB + ((ch + 1) & 1) * BUF;
#pragma unroll
      for (int it = 0; it < NLD; it++) {
        *reinterpret_cast<d2 *>(nA + (lrow + RPS * it) * LDK + 2 * lc2) = pa[it];
        *reinterpret_cast<d2 *>(nB + (lrow + RPS * it) * LDK + 2 * lc2) = pb[it];
      }
    }
    if (!DBUF) __syncthreads();
    const int nx = ch + (DBUF ? 2 : 1);
    if (nx < NCH) {  // next chunk -> registers, in flight while this chunk is multiplied
      const T *A = (NP == 2 && nx >= NB / KC) ? A1 : A0;
      const T *B = (NP == 2 && nx >= NB / KC) ? B1 : B0;
      const int k0 = (nx & (NB / KC - 1)) * KC;
#pragma unroll
      for (int it = 0; it < NLD; it++) {
        pa[it] = *reinterpret_cast<const d2 *>(A + (lrow + RPS * it) * NB + k0 + 2 * lc2);
        pb[it] = *reinterpret_cast<const d2 *>(B + (lrow + RPS * it) * NB + k0 + 2 * lc2);
      }
    }
    if (YACC) {
      if (tid < NB) {
        T ya = *yacc;
#pragma unroll
        for (int q = 0; q < KC; q++) ya += cB[tid * LDK + q] * bk[ch * KC + q];
        *yacc = ya;
      }
    }
#pragma unroll
    for (int kk = 0; kk < KC / 4; kk++) {
      T af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; m++) af[m] = cA[(wr + 16 * m + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int n = 0; n < 4; n++) bf[n] = cB[(wc + 16 * n + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = RT<T>::mfma(af[m], bf[n], acc[m][n]);
    }
  }
}

// Wave-private variant: every wave stages the 64 A rows and 64 B rows IT needs in its own LDS region, so the K loop has
// no workgroup barrier at all (each operand slice is loaded by the two waves that use it: 2x the L2->LDS traffic, which
// is not the limiter).  A chunk is 128 bytes of every row -- 16 doubles or 32 floats -- moved in 16-byte pieces, so both
// types issue the same loads / LDS writes per chunk against the same 4096 cycles of MFMA work (with 16-float chunks moved
// in 8-byte pieces the Float32 update ran at 57 % of its peak where the Float64 one reaches 70-75 %: the per-chunk
// overhead is fixed, the Float32 matrix instructions are twice as fast).  LDS: 4 waves x 128 rows x 144 bytes = 73.7 KB;
// row strides 18 doubles / 36 floats: 16-byte aligned, conflict-free operand reads.
template <typename T>
struct PV {
  static constexpr int VL = 16 / sizeof(T);                  // elements per 16-byte piece
  static constexpr int KC = 128 / sizeof(T);                 // chunk width
  static constexpr int LDK = KC + (sizeof(T) == 8 ? 2 : 4);  // LDS row stride (elements)
  typedef T vu __attribute__((ext_vector_type(16 / sizeof(T))));
};
template <typename T>
constexpr size_t gemm_priv_lds_bytes() { return (size_t)4 * 2 * 64 * PV<T>::LDK * sizeof(T); }
constexpr size_t GEMM_PRIV_LDS_ELEMS = gemm_priv_lds_bytes<double>() / sizeof(double);  // (Float64 micro-benchmarks)
template <typename T, int NP>
__device__ inline void tile_gemm_abt_priv(const T *__restrict__ A0, const T *__restrict__ B0,
                                          const T *__restrict__ A1, const T *__restrict__ B1, T *lds,
                                          typename RT<T>::v4 acc[4][4], const T *__restrict__ A2 = nullptr,
                                          const T *__restrict__ B2 = nullptr, const T *__restrict__ A3 = nullptr,
                                          const T *__restrict__ B3 = nullptr) {
  typedef typename PV<T>::vu vu;
  constexpr int PKC = PV<T>::KC, PLDK = PV<T>::LDK, VL = PV<T>::VL;
  const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3;  // (& 3: a workgroup of eight waves works on two tiles)
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  const int fr = lane & 15, fk = lane >> 4;
  T *sA = lds + wv * (2 * 64 * PLDK), *sB = sA + 64 * PLDK;
  constexpr int UPR = PKC / VL;     // 16-byte pieces per row of a chunk (8)
  constexpr int RPS = 64 / UPR;     // rows per step of the 64 lanes (8)
  constexpr int NLD = 64 / RPS;     // steps to cover 64 rows (8)
  const int lrow = lane / UPR, lc = (lane % UPR) * VL;
  vu pa[NLD], pb[NLD];
  const T *Ab = A0 + wr * NB, *Bb = B0 + wc * NB;
#pragma unroll
  for (int it = 0; it < NLD; it++) {
    pa[it] = *reinterpret_cast<const vu *>(Ab + (lrow + RPS * it) * NB + lc);
    pb[it] = *reinterpret_cast<const vu *>(Bb + (lrow + RPS * it) * NB + lc);
  }
  constexpr int NCH = NP * (NB / PKC);
  for (int ch = 0; ch < NCH; ch++) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NLD; it++) {
      *reinterpret_cast<vu *>(sA + (lrow + RPS * it) * PLDK + lc) = pa[it];
      *reinterpret_cast<vu *>(sB + (lrow + RPS * it) * PLDK + lc) = pb[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nx = ch + 1;
    if (nx < NCH) {
      const int pn = nx / (NB / PKC);  // panel of the next chunk (NP = 4: the K = 512 pass of the quad-update probe)
      const T *A = (pn == 0 ? A0 : pn == 1 ? A1 : pn == 2 ? A2 : A3) + wr * NB;
      const T *B = (pn == 0 ? B0 : pn == 1 ? B1 : pn == 2 ? B2 : B3) + wc * NB;
      const int k0 = (nx & (NB / PKC - 1)) * PKC;
#pragma unroll
      for (int it = 0; it < NLD; it++) {
        pa[it] = *reinterpret_cast<const vu *>(A + (lrow + RPS * it) * NB + k0 + lc);
        pb[it] = *reinterpret_cast<const vu *>(B + (lrow + RPS * it) * NB + k0 + lc);
      }
    }
#pragma unroll
    for (int kk = 0; kk < PKC / 4; kk++) {
      T af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; m++) af[m] = sA[(16 * m + fr) * PLDK + kk * 4 + fk];
#pragma unroll
      for (int n = 0; n < 4; n++) bf[n] = sB[(16 * n + fr) * PLDK + kk * 4 + fk];
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = RT<T>::mfma(af[m], bf[n], acc[m][n]);
    }
  }
}

// ---- row-split panel kernels (latency path) ----------------------------------------------------------------------------
// The panel solve and the one-column update have only (nt-k-1) tiles of work: one workgroup per tile leaves most CUs
// idle and takes a full 128x128x128 product (27-33 us) on the critical path of every panel.  Here a workgroup owns
// RS = 32 rows of a tile (4 waves x (32 rows x 32 columns) = 2x2 MFMA blocks each), so 4x as many workgroups run
// concurrently and each is ~4x shorter.
constexpr int RS = 32;
constexpr size_t RS_LDS_ELEMS = (size_t)(2 * (RS + NB) * LDK + 5 * NB);

// Latency is what these 32-row products cost (a chunk's MFMAs take ~0.4 us, a global load 1-2 us): ALL of the K range
// (8 chunks of KC) is requested up front -- 80 VGPRs of loads in flight, the latency is paid once -- and the chunks pass
// through a two-buffer LDS ring with ONE barrier each (a wave reaches barrier ch only after its MFMAs of chunk ch-1, so
// buffer (ch+1)&1 is free to be overwritten once barrier ch has been passed).
// LDS: sA = 2 x RS x LDK, sB = 2 x NB x LDK.
template <typename T, bool YACC>
__device__ inline void tile_gemm_rows(const T *__restrict__ A, const T *__restrict__ B, T *sA, T *sB,
                                      typename RT<T>::v4 acc[2][2], const T *__restrict__ bk, T *yacc) {
  BA_VT
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wc = wv * 32;
  const int fr = lane & 15, fk = lane >> 4;
  constexpr int UPR = KC / 2;
  const int lrow = tid / UPR, lc2 = tid % UPR;  // 256 threads cover 32 rows x (KC/2) 16-byte units
  constexpr int NCH = NB / KC;
  d2 pa[NCH], pb[NCH][4];
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    pa[ch] = *reinterpret_cast<const d2 *>(A + lrow * NB + ch * KC + 2 * lc2);
#pragma unroll
    for (int it = 0; it < 4; it++) pb[ch][it] = *reinterpret_cast<const d2 *>(B + (lrow + 32 * it) * NB + ch * KC + 2 * lc2);
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    T *cA = sA + (ch & 1) * RS * LDK, *cB = sB + (ch & 1) * NB * LDK;
    *reinterpret_cast<d2 *>(cA + lrow * LDK + 2 * lc2) = pa[ch];
#pragma unroll
    for (int it = 0; it < 4; it++) *reinterpret_cast<d2 *>(cB + (lrow + 32 * it) * LDK + 2 * lc2) = pb[ch][it];
    __syncthreads();
    if (YACC) {
      if (tid < NB) {
        T ya = *yacc;
#pragma unroll
        for (int q = 0; q < KC; q++) ya += cB[tid * LDK + q] * bk[ch * KC + q];
        *yacc = ya;
      }
    }
#pragma unroll
    for (int kk = 0; kk < KC / 4; kk++) {
      T af[2], bf[2];
#pragma unroll
      for (int m = 0; m < 2; m++) af[m] = cA[(16 * m + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int n = 0; n < 2; n++) bf[n] = cB[(wc + 16 * n + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) acc[m][n] = RT<T>::mfma(af[m], bf[n], acc[m][n]);
    }
  }
  __syncthreads();  // the ring (and whatever the caller keeps beside it) may be rewritten from here on
}

// rows [32 rq, 32 rq + 32) of X_i = S_ik Linv_k' -> V_i, S_ik = X_i D_k^-1; FWD: y_k and b_i -= L_ik y_k ride along.
// grid = 4 (nt-k-1): i = k + 1 + blockIdx.x / 4, rq = blockIdx.x % 4
template <typename T, bool FWD>
__device__ __forceinline__ void ldl_trsm_rs_body(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv_k,
                                        const T *__restrict__ D_k, T *__restrict__ V, int k, T *__restrict__ b, T *__restrict__ y,
                                        const int *__restrict__ rows, const int bid) {
  BA_VT
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  T *sA = lds, *sB = lds + 2 * RS * LDK, *ysh = lds + 2 * (RS + NB) * LDK, *red = ysh + NB;  // red: 4 x 32
  // rows (block-sparse S): the tile rows of this panel's pattern, ascending; null: every tile row below the diagonal tile
  const int i = rows ? rows[bid >> 2] : k + 1 + (bid >> 2), r0 = (bid & 3) * RS;
  T *Sik = S + tix(co, i, k) * NB * NB + r0 * NB;
  T *Vi = V + (int64_t)i * NB * NB + r0 * NB;
  typename RT<T>::v4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = (d4){0, 0, 0, 0};
  T yacc = 0;
  tile_gemm_rows<T, FWD>(Sik, Linv_k, sA, sB, acc, FWD ? b + (int64_t)k * NB : nullptr, &yacc);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wc = wv * 32;
  if (FWD) {
    if (tid < NB) {
      ysh[tid] = yacc;
      if (bid == 0) y[(int64_t)k * NB + tid] = yacc;
    }
    __syncthreads();
  }
  T part[2][4];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int g = 0; g < 4; g++) part[m][g] = 0;
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int col = wc + 16 * n + (lane & 15);
    const T inv_d = (T)1 / D_k[col];
    const T wcol = FWD ? ysh[col] * inv_d : 0.0;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * m + RT<T>::row(lane, g);
        const T xv = acc[m][n][g];
        Vi[row * NB + col] = xv;
        Sik[row * NB + col] = xv * inv_d;
        if (FWD) part[m][g] += xv * wcol;
      }
  }
  if (FWD) {
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        T v = part[m][g];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if ((lane & 15) == 0) red[wv * RS + 16 * m + RT<T>::row(lane, g)] = v;
      }
    __syncthreads();
    if (tid < RS) b[(int64_t)i * NB + r0 + tid] -= ((red[tid] + red[RS + tid]) + red[2 * RS + tid]) + red[3 * RS + tid];
  }
}

template <typename T, bool FWD>
__global__ __launch_bounds__(256) void k_ldl_trsm_rs(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv_k,
                                                      const T *__restrict__ D_k, T *__restrict__ V, int k,
                                                      T *__restrict__ b, T *__restrict__ y,
                                                      const int *__restrict__ rows = nullptr) {
  ldl_trsm_rs_body<T, FWD>(S, co, Linv_k, D_k, V, k, b, y, rows, (int)blockIdx.x);
}

// ---- two runs per launch -------------------------------------------------------------------------------------------------
// Two independent runs of tile column pairs (TilePattern::a_clean / b_clean: a profile eliminated from both ends) advance
// TOGETHER: every kernel of the panel chain is launched once for both runs -- the first n0 workgroups work on run A's panel,
// the others on run B's -- so a step of two pairs costs the launches (and the latency: these kernels leave the chip almost
// empty) of one.  What a run needs per panel: its tile column k, the inverted diagonal tile and pivots of that column, its
// panel buffer, its row list.  (Two streams, one per run, were tried first: the replayed graph ran the two branches one after
// the other, kernel trace profiles/r04_d_trace_final_two_streams_timeline.txt.)
template <typename T>
struct RunPanel {
  const T *Linv_k, *D_k;
  T *V;
  int k;
  const int *rows;
};
template <typename T, bool FWD>
__global__ __launch_bounds__(256) void k_ldl_trsm_rs2(T *__restrict__ S, const int64_t *__restrict__ co, RunPanel<T> a, RunPanel<T> c, int n0,
                                                       T *__restrict__ b, T *__restrict__ y) {
  const bool second = (int)blockIdx.x >= n0;
  ldl_trsm_rs_body<T, FWD>(S, co, second ? c.Linv_k : a.Linv_k, second ? c.D_k : a.D_k, second ? c.V : a.V, second ? c.k : a.k, b, y,
                           second ? c.rows : a.rows, (int)blockIdx.x - (second ? n0 : 0));
}

// rows [32 rq, 32 rq + 32) of S_{i,k+1} -= V0_i L_{k+1,k}'   (grid = 4 (nt-k-1))
template <typename T>
__device__ __forceinline__ void ldl_col_rs_body(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0, int k,
                                       const int *__restrict__ rows, const int bid) {
  BA_VT
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  T *sA = lds, *sB = lds + 2 * RS * LDK;
  const int i = rows ? rows[bid >> 2] : k + 1 + (bid >> 2), r0 = (bid & 3) * RS;
  typename RT<T>::v4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_rows<T, false>(V0 + (int64_t)i * NB * NB + r0 * NB, S + tix(co, k + 1, k) * NB * NB, sA, sB, acc, nullptr, nullptr);
  const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) * 32;
  T *Sij = S + tix(co, i, k + 1) * NB * NB + r0 * NB;
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int col = wc + 16 * n + (lane & 15);
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * m + RT<T>::row(lane, g);
        Sij[row * NB + col] -= acc[m][n][g];
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_ldl_col_rs(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0, int k,
                                                     const int *__restrict__ rows = nullptr) {
  ldl_col_rs_body<T>(S, co, V0, k, rows, (int)blockIdx.x);
}
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_col_rs2(T *__restrict__ S, const int64_t *__restrict__ co, RunPanel<T> a, RunPanel<T> c, int n0) {
  const bool second = (int)blockIdx.x >= n0;
  ldl_col_rs_body<T>(S, co, second ? c.V : a.V, second ? c.k : a.k, second ? c.rows : a.rows, (int)blockIdx.x - (second ? n0 : 0));
}

// rows [32 rq, 32 rq + 32) of one tile of the pair update, S_ij -= V0_i L_jk' + V1_i L_{j,k+1}' (grid = 4 nblk): the
// row-split form of k_ldl_update<1> for SHORT updates.  A tile of the big kernel is bound by the matrix pipe of the four
// waves that own it -- 1024 MFMAs per wave, ~30 us, however empty the chip is -- so an update of a few hundred tiles (the
// block-sparse schedule's, Dubrovnik-356's, LadyBug-49's, the tail of Venice's) takes ~45 us for a fraction of a round.  Here
// a tile is four workgroups of 32 rows (256 MFMAs per wave), three of them resident per CU.  The same tile enumeration as
// k_ldl_update (triangular index, optional row list of the block-sparse pattern); the whole K range of each panel is
// requested up front (tile_gemm_rows).  (Tried in round 3 and dropped for these short updates: running the rest of the update
// on a second stream beside the next diagonal tile -- the fork / join costs what it hides: 19.17 against 18.95 ms per LM
// iteration on the Venice shape with 24 % block fill.)
// lead_len > 0 (look-ahead of the block-sparse schedule): only the tiles of the first one or two COLUMNS of the list are
// updated -- (rows[ii], rows[0]), ii = 0 .. lead_len-1, then (rows[ii], rows[1]), ii = 1 .. lead_len-1 -- what the next pair's
// panel chain needs; the rest of the update runs beside that chain (dense_ldl_factor_sparse).
template <typename T>
__device__ __forceinline__ void ldl_update_rs_body(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0,
                                          const T *__restrict__ V1, int k, int base, int nblk, const int *__restrict__ rows,
                                          int lead_len, const int bid) {
  BA_VT
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  T *sA = lds, *sB = lds + 2 * RS * LDK;
  const int t = bid >> 2, r0 = (bid & 3) * RS;
  if (t >= nblk) return;
  int ii, jj;
  if (lead_len > 0) {
    jj = t < lead_len ? 0 : 1;
    ii = t < lead_len ? t : t - lead_len + 1;
  } else {
    ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
    while (ii * (ii + 1) / 2 > t) ii--;
    jj = t - ii * (ii + 1) / 2;
  }
  const int i = rows ? rows[ii] : base + ii, j = rows ? rows[jj] : base + jj;
  typename RT<T>::v4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_rows<T, false>(V0 + (int64_t)i * NB * NB + r0 * NB, S + tix(co, j, k) * NB * NB, sA, sB, acc, nullptr, nullptr);
  tile_gemm_rows<T, false>(V1 + (int64_t)i * NB * NB + r0 * NB, S + tix(co, j, k + 1) * NB * NB, sA, sB, acc, nullptr, nullptr);
  const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) * 32;
  T *Sij = S + tix(co, i, j) * NB * NB + r0 * NB;
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int col = wc + 16 * n + (lane & 15);
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * m + RT<T>::row(lane, g);
        Sij[row * NB + col] -= acc[m][n][g];
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_ldl_update_rs(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0,
                                                        const T *__restrict__ V1, int k, int base, int nblk,
                                                        const int *__restrict__ rows, int lead_len = 0) {
  ldl_update_rs_body<T>(S, co, V0, V1, k, base, nblk, rows, lead_len, (int)blockIdx.x);
}
// the pair updates of two runs in one launch (V: the run's first panel buffer, the second follows it at a panel's distance)
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_update_rs2(T *__restrict__ S, const int64_t *__restrict__ co, RunPanel<T> a, int nblk_a,
                                                         RunPanel<T> c, int nblk_c, int64_t panel, int lead_a = 0, int lead_c = 0) {
  // lead_a / lead_c > 0: the run's launch part is the lead strip of its look-ahead (see k_ldl_update_rs)
  const bool second = (int)blockIdx.x >= 4 * nblk_a;
  const RunPanel<T> &r = second ? c : a;
  ldl_update_rs_body<T>(S, co, r.V, r.V + panel, r.k, r.k + 2, second ? nblk_c : nblk_a, r.rows, second ? lead_c : lead_a,
                        (int)blockIdx.x - (second ? 4 * nblk_a : 0));
}

// ---- fused panel-pair kernels (pair schedule) ------------------------------------------------------------------------------
// The panel chain of a pair (k, k+1) -- diag(k), panel solve of column k, update of column k+1, diag(k+1), panel solve of
// column k+1 -- has only TWO steps that are inherently sequential workgroup-sized jobs, the two diagonal tiles, and the
// second one needs nothing of panel k but tile row k+1.  k_ldl_pairdiag does everything that involves only the three
// tiles (k,k), (k+1,k), (k+1,k+1) in ONE workgroup:
//     diag(k);  X = S_{k+1,k} Linv_k';  V0_{k+1} = X, L_{k+1,k} = X D_k^-1;  S_{k+1,k+1} -= X L_{k+1,k}';  diag(k+1)
// (+ the forward substitution of rows k, k+1), so that it can be hoisted as a whole beside the previous pair's trailing
// update (it needs that update's first three tiles only).  What is left on the critical path is ONE row-parallel kernel,
// k_ldl_pairtrsm: for every 32-row slice of the tile rows i >= k+2
//     X0 = S_ik Linv_k';  S_{i,k+1} -= X0 L_{k+1,k}';  X1 = S_{i,k+1} Linv_{k+1}'      (X0, then the updated slice, stay in LDS)
// and the forward substitution of its rows: b_i -= L_ik y_k + L_{i,k+1} y_{k+1}.
constexpr int LDX = NB + 4;  // row stride of the 32 x 128 slice kept in LDS between the stages (264 dwords = 8 mod 64)
constexpr size_t PT_LDS_ELEMS = (size_t)(RS * LDX + 2 * NB * LDK + 2 * NB + 4 * RS);  // 73.7 KB in f64: two workgroups per CU

// C(32 x 128, distributed like tile_gemm_rows) = A B' with the 32 x 128 A slice already in LDS (sX, stride LDX) and the B
// tile streamed from global memory in chunks of KC through sB
template <typename T>
__device__ inline void tile_gemm_rows_ldsA(const T *sX, const T *__restrict__ B, T *sB, typename RT<T>::v4 acc[2][2]) {
  BA_VT
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wc = wv * 32;
  const int fr = lane & 15, fk = lane >> 4;
  constexpr int UPR = KC / 2;
  const int lrow = tid / UPR, lc2 = tid % UPR;
  constexpr int NCH = NB / KC;
  d2 pb[NCH][4];  // the whole B tile is requested up front (see tile_gemm_rows)
#pragma unroll
  for (int ch = 0; ch < NCH; ch++)
#pragma unroll
    for (int it = 0; it < 4; it++) pb[ch][it] = *reinterpret_cast<const d2 *>(B + (lrow + 32 * it) * NB + ch * KC + 2 * lc2);
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    T *cB = sB + (ch & 1) * NB * LDK;
#pragma unroll
    for (int it = 0; it < 4; it++) *reinterpret_cast<d2 *>(cB + (lrow + 32 * it) * LDK + 2 * lc2) = pb[ch][it];
    __syncthreads();  // also orders the caller's writes of sX before the first chunk's reads
#pragma unroll
    for (int kk = 0; kk < KC / 4; kk++) {
      T af[2], bf[2];
#pragma unroll
      for (int m = 0; m < 2; m++) af[m] = sX[(16 * m + fr) * LDX + ch * KC + kk * 4 + fk];
#pragma unroll
      for (int n = 0; n < 2; n++) bf[n] = cB[(wc + 16 * n + fr) * LDK + kk * 4 + fk];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) acc[m][n] = RT<T>::mfma(af[m], bf[n], acc[m][n]);
    }
  }
  __syncthreads();  // every wave has finished reading sX and the ring
}

// grid = 4 (nt-k-2): i = k + 2 + blockIdx.x / 4, rows [32 rq, 32 rq + 32), rq = blockIdx.x % 4.  y: y_k | y_{k+1} at y + k NB.
template <typename T, bool FWD>
__global__ __launch_bounds__(256) void k_ldl_pairtrsm(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                                       const T *__restrict__ D, T *__restrict__ V0, T *__restrict__ V1, int k,
                                                       T *__restrict__ b, const T *__restrict__ y) {
  BA_VT
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  T *sX = lds, *sB = sX + RS * LDX, *ysh = sB + 2 * NB * LDK, *red = ysh + 2 * NB;  // red: 4 x 32
  const int i = k + 2 + (blockIdx.x >> 2), r0 = (blockIdx.x & 3) * RS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wc = wv * 32;
  T *Sik = S + tix(co, i, k) * NB * NB + r0 * NB;
  T *Sik1 = S + tix(co, i, k + 1) * NB * NB + r0 * NB;
  T *Vi0 = V0 + (int64_t)i * NB * NB + r0 * NB, *Vi1 = V1 + (int64_t)i * NB * NB + r0 * NB;
  const T *Dk = D + (int64_t)k * NB, *Dk1 = Dk + NB;
  if (FWD) ysh[tid] = y[(int64_t)k * NB + tid];  // y_k | y_{k+1} (256 values); ordered by the barriers of the first product
  typename RT<T>::v4 acc[2][2];
  T part[2][4];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int g = 0; g < 4; g++) part[m][g] = 0;
  // ---- stage 1: X0 = S_ik Linv_k'
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = (d4){0, 0, 0, 0};
  {  // the 32 x 128 slice of S_ik goes to LDS whole (16 loads of 16 bytes per thread in flight)
    d2 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const d2 *>(Sik + 2 * (q * 256 + tid));
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int idx = 2 * (q * 256 + tid);
      *reinterpret_cast<d2 *>(sX + (idx >> 7) * LDX + (idx & (NB - 1))) = v[q];
    }
  }
  tile_gemm_rows_ldsA<T>(sX, Linv + (int64_t)k * NB * NB, sB, acc);
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int col = wc + 16 * n + (lane & 15);
    const T inv_d = (T)1 / Dk[col];
    const T wcol = FWD ? ysh[col] * inv_d : (T)0;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * m + RT<T>::row(lane, g);
        const T xv = acc[m][n][g];
        Vi0[row * NB + col] = xv;
        Sik[row * NB + col] = xv * inv_d;
        sX[row * LDX + col] = xv;
        if (FWD) part[m][g] += xv * wcol;
      }
  }
  // ---- stage 2: C = S_{i,k+1} - X0 L_{k+1,k}'   (the barriers inside the product order the sX writes above)
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_rows_ldsA<T>(sX, S + tix(co, k + 1, k) * NB * NB, sB, acc);  // returns after every wave has finished reading X0
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int col = wc + 16 * n + (lane & 15);
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * m + RT<T>::row(lane, g);
        sX[row * LDX + col] = Sik1[row * NB + col] - acc[m][n][g];
      }
  }
  // ---- stage 3: X1 = C Linv_{k+1}'
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_rows_ldsA<T>(sX, Linv + (int64_t)(k + 1) * NB * NB, sB, acc);
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const int col = wc + 16 * n + (lane & 15);
    const T inv_d = (T)1 / Dk1[col];
    const T wcol = FWD ? ysh[NB + col] * inv_d : (T)0;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * m + RT<T>::row(lane, g);
        const T xv = acc[m][n][g];
        Vi1[row * NB + col] = xv;
        Sik1[row * NB + col] = xv * inv_d;
        if (FWD) part[m][g] += xv * wcol;
      }
  }
  if (FWD) {  // b_i -= L_ik y_k + L_{i,k+1} y_{k+1} for this slice's rows: fixed tree (16 lanes, then the 4 waves)
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        T v = part[m][g];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if ((lane & 15) == 0) red[wv * RS + 16 * m + RT<T>::row(lane, g)] = v;
      }
    __syncthreads();
    if (tid < RS) b[(int64_t)i * NB + r0 + tid] -= ((red[tid] + red[RS + tid]) + red[2 * RS + tid]) + red[3 * RS + tid];
  }
}

// one workgroup: the three tiles (k,k), (k+1,k), (k+1,k+1) of a pair (see above).  wait_ready / need: hoisted launch.
template <typename T, bool FWD>
__global__ __launch_bounds__(256) void k_ldl_pairdiag(T *__restrict__ S, const int64_t *__restrict__ co, T *__restrict__ Linv,
                                                       T *__restrict__ D, T *__restrict__ V0, int k, int nt,
                                                       int *__restrict__ flag, const int *__restrict__ wait_ready, int need,
                                                       T *__restrict__ b, T *__restrict__ y) {
  BA_VT
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *sm = reinterpret_cast<T *>(smraw);
  if (wait_ready && !hoisted_wait(wait_ready, need, flag)) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  T *Skk = S + tix(co, k, k) * NB * NB, *Lk = Linv + (int64_t)k * NB * NB, *Dk = D + (int64_t)k * NB;
  diag_tile<T, 4>(Skk, Lk, Dk, flag, nullptr, sm);
  __threadfence();  // this workgroup re-reads what it has just written (Linv_k now, L_{k+1,k} and V0_{k+1} below) from memory
  __syncthreads();
  T *ysh = sm + 2 * NB * LDK;  // behind the product's staging area
  if (FWD) {  // y_k = Linv_k b_k  (unit lower triangular: columns <= row)
    if (tid < NB) {
      T sacc = 0;
      for (int c = 0; c <= tid; c++) sacc += Lk[tid * NB + c] * b[(int64_t)k * NB + c];
      ysh[tid] = sacc;
      y[(int64_t)k * NB + tid] = sacc;
    }
    __syncthreads();
  }
  if (k + 1 >= nt) return;
  T *S10 = S + tix(co, k + 1, k) * NB * NB, *S11 = S + tix(co, k + 1, k + 1) * NB * NB;
  T *V10 = V0 + (int64_t)(k + 1) * NB * NB;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  typename RT<T>::v4 acc[4][4];
  // X = S_{k+1,k} Linv_k' -> V0_{k+1}, L_{k+1,k} = X D_k^-1
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_abt<T, 1>(S10, Lk, nullptr, nullptr, sm, sm + NB * LDK, acc);
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int col = wc + 16 * n + (lane & 15);
    const T inv_d = (T)1 / Dk[col];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = wr + 16 * m + RT<T>::row(lane, g);
        const T xv = acc[m][n][g];
        V10[row * NB + col] = xv;
        S10[row * NB + col] = xv * inv_d;
      }
  }
  __threadfence();
  __syncthreads();
  if (FWD) {  // b_{k+1} -= L_{k+1,k} y_k
    if (tid < NB) {
      T sacc = 0;
      for (int c = 0; c < NB; c++) sacc += S10[tid * NB + c] * ysh[c];
      b[(int64_t)(k + 1) * NB + tid] -= sacc;
    }
  }
  // S_{k+1,k+1} -= X L_{k+1,k}'
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) acc[m][n] = (d4){0, 0, 0, 0};
  tile_gemm_abt<T, 1>(V10, S10, nullptr, nullptr, sm, sm + NB * LDK, acc);
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int col = wc + 16 * n + (lane & 15);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = wr + 16 * m + RT<T>::row(lane, g);
        S11[row * NB + col] -= acc[m][n][g];
      }
  }
  __threadfence();
  __syncthreads();
  T *Lk1 = Lk + NB * NB, *Dk1 = Dk + NB;
  diag_tile<T, 4>(S11, Lk1, Dk1, flag, nullptr, sm);
  if (FWD) {  // y_{k+1} = Linv_{k+1} b_{k+1}
    __threadfence();
    __syncthreads();
    if (tid < NB) {
      T sacc = 0;
      for (int c = 0; c <= tid; c++) sacc += Lk1[tid * NB + c] * b[(int64_t)(k + 1) * NB + c];
      y[(int64_t)(k + 1) * NB + tid] = sacc;
    }
  }
}

// one tile of the pair update: S_ij -= V0_i L_jk' + V1_i L_{j,k+1}' (K = 256) by the four waves of a workgroup
template <typename T>
__device__ __forceinline__ void ldl_update_tile(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0, const T *__restrict__ V1,
                                       int k, int i, int j, T *lds, const T *__restrict__ Lp0, const T *__restrict__ Lp1) {
  BA_VT
  T *Sij = S + tix(co, i, j) * NB * NB;
  typename RT<T>::v4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) acc[m][n] = (d4){0, 0, 0, 0};
  // Lp0 / Lp1 (distributed factorisation with per-rank ownership of S): the L tiles of the two panels come from the panel
  // buffers the broadcast filled (tile row j at Lp + j NB^2) -- a rank holds only its own tile columns of S
  tile_gemm_abt_priv<T, 2>(V0 + (int64_t)i * NB * NB, Lp0 ? Lp0 + (int64_t)j * NB * NB : S + tix(co, j, k) * NB * NB,
                        V1 + (int64_t)i * NB * NB, Lp1 ? Lp1 + (int64_t)j * NB * NB : S + tix(co, j, k + 1) * NB * NB, lds, acc);
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));  // keep the epilogue's address arithmetic out of the main loop's live ranges
  const int lane = tid2 & 63, wv = (tid2 >> 6) & 3;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  // epilogue: the 64 values of a lane are read-modify-written in two batches of 32 so that 32 loads are in flight at once
  T *cbase = Sij + wr * NB + wc + (lane & 15);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    T cv[2][4][4];
#pragma unroll
    for (int n2 = 0; n2 < 2; n2++)
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int g = 0; g < 4; g++) cv[n2][m][g] = NT_C ? __builtin_nontemporal_load(&cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)]) : cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)];
#pragma unroll
    for (int n2 = 0; n2 < 2; n2++)
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const T nv = cv[n2][m][g] - acc[m][2 * h + n2][g];
          if (NT_C) __builtin_nontemporal_store(nv, &cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)]);
          else cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)] = nv;
        }
  }
}

// Bulk trailing update, two panels per pass:  S_ij -= V0_i L_jk' + V1_i L_{j,k+1}'  for the lower-triangular tile pairs
// base <= j <= i (K = 256): the trailing matrix is read and written once per TWO panels, which halves its HBM traffic per
// flop.  (MODE is kept as a template parameter; only MODE 1 exists.  The probe variants this kernel was tuned with --
// store-only epilogue, L2-resident operands, workgroup-shared staging, LDS-DMA staging, K = 512, accumulators started from
// -C -- live in csrc/bench/ba_bench_ldl.hip as k_ldl_update_probe, built into tools/libba_bench.so only.)
// Where the remaining ~25 % are NOT: the K loop's MFMA + LDS-read stream alone runs at 78.0 TFLOP/s (k_mfma_probe, modes
// 0 and 1: the operand reads are free); the two workgroups of a CU having their prologue / epilogue at the same time
// (delaying every CU's second workgroup by 10 / 20 / 34 us changed nothing: 59.6-59.9); the read-modify-write epilogue
// and the operands' home (probe variants 1, 2).  What is left is the staging inside the loop: every wave pulls its own A
// and B slices, 16 KB per 64 MFMAs, ~9.7 TB/s of L2 -> LDS traffic chip-wide, with a wait for it at every chunk boundary.
// OWN (distributed factorisation): the launch covers only the tile columns this rank owns, listed ascending in own_cols
// with own_pref[m] = number of tiles in the columns before own_cols[m]; the columns >= base start at index m0.
template <typename T, int MODE, bool OWN = false>
__global__ __launch_bounds__(256, 2) void k_ldl_update(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0,
                                                        const T *__restrict__ V1, int k, int base, int nt,
                                                        int nblk, int *__restrict__ ready,
                                                        const int *__restrict__ own_cols = nullptr,
                                                        const int64_t *__restrict__ own_pref = nullptr, int m0 = 0,
                                                        int m_end = 0, int ready_tiles = 1, const int *__restrict__ rows = nullptr,
                                                        const T *__restrict__ Lp0 = nullptr, const T *__restrict__ Lp1 = nullptr,
                                                        const int2 *__restrict__ tlist = nullptr, int blocked = TSB) {
  // tlist (block-sparse S on several ranks): the tiles (i, j) of this launch, listed (the pattern's tiles of the pair's update
  // in the tile columns this rank owns)
  BA_VT
  static_assert(MODE == 1, "only the pair update is a tile-per-workgroup kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  int i, j, tsel;
  {
    // chunked block -> XCD map: blocks b, b+8, ... share an XCD (round-robin dispatch); give each XCD a contiguous
    // range of tile rows so that V_i stays in its L2 (speed only)
    const int per = (nblk + 7) / 8;
    int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= nblk) return;
    tsel = t;
    if (tlist) {
      i = tlist[t].x;
      j = tlist[t].y;
    } else if (OWN) {
      const int64_t tt = t + own_pref[m0];
      int lo = m0, hi = m_end;  // largest m with own_pref[m] <= tt
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (own_pref[mid] <= tt) lo = mid;
        else hi = mid;
      }
      j = own_cols[lo];
      i = j + (int)(tt - own_pref[lo]);
    } else {
      // the lower triangle in super-blocks of 8 x 8 tiles (tri_blocked, ba_internal.h): the ~64 workgroups an XCD runs at a
      // time share 8 + 8 row panels instead of 1 + 64 (row-major enumeration: every tile its own B panel from beyond the L2)
      int m_rows = (int)((sqrt(8.0 * (double)nblk + 1.0) - 1.0) * 0.5);  // nblk = m_rows (m_rows + 1) / 2 tiles
      while ((m_rows + 1) * (m_rows + 2) / 2 <= nblk) m_rows++;
      while (m_rows * (m_rows + 1) / 2 > nblk) m_rows--;
      int ii, jj;
      if (blocked) {
        tri_blocked(t, m_rows, &ii, &jj, blocked);
      } else {
        ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
        while (ii * (ii + 1) / 2 > t) ii--;
        jj = t - ii * (ii + 1) / 2;
      }
      // rows (block-sparse S): the pair's pattern, ascending -- tile (rows[ii], rows[jj]) instead of (base + ii, base + jj)
      i = rows ? rows[ii] : base + ii;
      j = rows ? rows[jj] : base + jj;
    }
  }
  ldl_update_tile<T>(S, co, V0, V1, k, i, j, lds, Lp0, Lp1);
  // tiles 0 .. ready_tiles-1 are (base,base) [, (base+1,base), (base+1,base+1)]: what the next pair's hoisted diagonal
  // kernel waits for.  Each tells it so once its tile is final.
  if (ready && tsel < ready_tiles) {
    __threadfence();  // every thread's stores, agent scope (written back past this XCD's L2)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same update for the look-ahead of the block-sparse schedule (dense_ldl_factor_sparse): the "rest" of a pair's
// trailing update runs BESIDE the next pair's panel chain, whose diagonal-tile kernel needs a whole CU to itself (153 KB of
// LDS in Float64) -- beside a launch that fills the chip it would wait until some CU has drained, i.e. until the update is
// over.  So this form keeps to a part of the chip: a fixed number of workgroups (grid), each asking for more than half a
// CU's LDS so that no two share a CU, walking the tiles t = block, block + grid, ...; the CUs it leaves alone take the
// chain.  Same tile routine, same enumeration (triangular index over the row list): same bits as k_ldl_update.
// Eight waves per workgroup: its two halves walk the tiles independently (nothing in the tile routine synchronises beyond the
// wave), so that a CU runs two waves per SIMD as under k_ldl_update -- with four waves (first version) a CU of the rest did
// 60 % of the work it does under the full launch.
template <typename T>
constexpr size_t part_lds_bytes() { return 2 * gemm_priv_lds_bytes<T>(); }  // 147 KB: one workgroup per CU
template <typename T>
__global__ __launch_bounds__(512) void k_ldl_update_part(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0,
                                                          const T *__restrict__ V1, int k, int nblk, const int *__restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  T *lds = reinterpret_cast<T *>(smraw + half * gemm_priv_lds_bytes<T>());
  for (int t = 2 * blockIdx.x + half; t < nblk; t += 2 * gridDim.x) {
    int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
    while (ii * (ii + 1) / 2 > t) ii--;
    const int jj = t - ii * (ii + 1) / 2;
    ldl_update_tile<T>(S, co, V0, V1, k, rows[ii], rows[jj], lds, nullptr, nullptr);
  }
}

// k_ldl_update_part for two runs: the rests of both runs' updates walked by one set of persistent workgroups
template <typename T>
__global__ __launch_bounds__(512) void k_ldl_update_part2(T *__restrict__ S, const int64_t *__restrict__ co, RunPanel<T> a, int nblk_a,
                                                           RunPanel<T> c, int nblk_c, int64_t panel) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  T *lds = reinterpret_cast<T *>(smraw + half * gemm_priv_lds_bytes<T>());
  for (int tt = 2 * blockIdx.x + half; tt < nblk_a + nblk_c; tt += 2 * gridDim.x) {
    const bool second = tt >= nblk_a;
    const RunPanel<T> &r = second ? c : a;
    const int t = tt - (second ? nblk_a : 0);
    int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
    while (ii * (ii + 1) / 2 > t) ii--;
    const int jj = t - ii * (ii + 1) / 2;
    ldl_update_tile<T>(S, co, r.V, r.V + panel, r.k, r.rows[ii], r.rows[jj], lds, nullptr, nullptr);
  }
}

// the pair updates of two runs in one launch, tile per workgroup (k_ldl_update's tile routine and enumeration over the run's
// row list; the chunked block -> XCD map inside each run's part of the grid)
template <typename T>
__global__ __launch_bounds__(256, 2) void k_ldl_update2(T *__restrict__ S, const int64_t *__restrict__ co, RunPanel<T> a, int nblk_a,
                                                         RunPanel<T> c, int nblk_c, int64_t panel) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  const int ga = ((nblk_a + 7) / 8) * 8;
  const bool second = (int)blockIdx.x >= ga;
  const RunPanel<T> &r = second ? c : a;
  const int nblk = second ? nblk_c : nblk_a, bid = (int)blockIdx.x - (second ? ga : 0);
  const int per = (nblk + 7) / 8;
  const int t = (bid & 7) * per + (bid >> 3);
  if (t >= nblk) return;
  int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
  while (ii * (ii + 1) / 2 > t) ii--;
  const int jj = t - ii * (ii + 1) / 2;
  ldl_update_tile<T>(S, co, r.V, r.V + panel, r.k, r.rows[ii], r.rows[jj], lds, nullptr, nullptr);
}

// ---- distributed factorisation helpers -----------------------------------------------------------------------------------
// A rank that received the panel V = L D of tile column k from its owner rebuilds L_ik = V_i D_k^-1 in its own copy of S
// (rows i0..nt-1), with the owner's arithmetic (k_ldl_trsm_rs: xv * (1 / d)), so that every rank holds the same bits.
template <typename T>
__global__ __launch_bounds__(256) void k_ldl_scale_panel(T *__restrict__ Lcol, const T *__restrict__ V, const T *__restrict__ D_k, int i0,
                                                          const int *__restrict__ rows = nullptr) {
  // Lcol: where the L tiles of this tile column live, tile row i at Lcol + i NB^2 (a column of S is contiguous: S + (co[k] - k)
  // NB^2; with per-rank ownership of S: a panel buffer)
  BA_VT
  const int i = rows ? rows[blockIdx.x] : i0 + blockIdx.x;  // rows: the tile rows of the panel's pattern (block-sparse S)
  const T *Vi = V + (int64_t)i * NB * NB;
  T *Sik = Lcol + (int64_t)i * NB * NB;
  const int c2 = threadIdx.x & 63;  // this thread's column pair (2 c2, 2 c2 + 1), the same in every row it touches
  const T inv0 = (T)1 / D_k[2 * c2], inv1 = (T)1 / D_k[2 * c2 + 1];
  for (int r = threadIdx.x >> 6; r < NB; r += 4) {
    d2 v = *reinterpret_cast<const d2 *>(Vi + r * NB + 2 * c2);
    v.x *= inv0;
    v.y *= inv1;
    *reinterpret_cast<d2 *>(Sik + r * NB + 2 * c2) = v;
  }
}

__global__ void k_flag_to_double(const int *flag, double *out) { out[0] = (double)(flag[0] != 0); }
__global__ void k_double_to_flag(const double *in, int *flag) { if (in[0] != 0.0) flag[0] = 1; }

// ---- triangular solves -------------------------------------------------------------------------------------------------
template <typename T>
__device__ inline T wsum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// forward step k: y_k = Linv_k b_k (every workgroup recomputes it; block 0 stores it), then b_i -= L_ik y_k, i > k.
template <typename T>
__global__ __launch_bounds__(256) void k_fwd_step(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                                   T *__restrict__ b, T *__restrict__ y, int k, const int *__restrict__ rows = nullptr,
                                                   const T *__restrict__ Lcol = nullptr) {
  BA_VT
  __shared__ T yk[NB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const T *Lk = Linv + (int64_t)k * NB * NB;
  const d2 bk = *reinterpret_cast<const d2 *>(b + (int64_t)k * NB + 2 * lane);
  for (int rr = 0; rr < 32; rr++) {
    int row = wv * 32 + rr;
    d2 l = *reinterpret_cast<const d2 *>(Lk + row * NB + 2 * lane);
    T s = wsum(l.x * bk.x + l.y * bk.y);
    if (lane == 0) yk[row] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < NB) y[(int64_t)k * NB + tid] = yk[tid];
    return;
  }
  const int i = rows ? rows[blockIdx.x - 1] : k + blockIdx.x;
  const T *Lik = Lcol ? Lcol + (int64_t)i * NB * NB : S + tix(co, i, k) * NB * NB;  // Lcol: panel buffer (per-rank ownership of S)
  const T y0 = yk[2 * lane], y1 = yk[2 * lane + 1];
  for (int rr = 0; rr < 32; rr++) {
    int row = wv * 32 + rr;
    d2 l = *reinterpret_cast<const d2 *>(Lik + row * NB + 2 * lane);
    T s = wsum(l.x * y0 + l.y * y1);
    if (lane == 0) b[(int64_t)i * NB + row] -= s;
  }
}

// backward step k: x_k = Linv_k' z_k with z = y / D (every workgroup recomputes it; block 0 stores it into b_k),
// then y_j -= D_j (L_kj' x_k) ... expressed on z: z_j -= L_kj' x_k, i.e. y_j -= D_j * (L_kj' x_k), j < k.
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_step(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                                   const T *__restrict__ D, T *__restrict__ y,
                                                   T *__restrict__ x, int k, const int *__restrict__ cols = nullptr) {
  BA_VT
  __shared__ T zk[NB], xk[NB], part[2][NB];
  const int tid = threadIdx.x;
  const int c = tid & (NB - 1), half = tid >> 7;
  if (tid < NB) zk[tid] = y[(int64_t)k * NB + tid] / D[(int64_t)k * NB + tid];
  __syncthreads();
  {
    const T *Lk = Linv + (int64_t)k * NB * NB;
    T s = 0;
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
  const int j = cols ? cols[blockIdx.x - 1] : blockIdx.x - 1;  // 0 .. k-1 (block-sparse S: the tile columns of row k's pattern)
  const T *Lkj = S + tix(co, k, j) * NB * NB;
  T s = 0;
  for (int r = half * 64; r < half * 64 + 64; r++) s += Lkj[r * NB + c] * xk[r];
  part[half][c] = s;
  __syncthreads();
  if (tid < NB) y[(int64_t)j * NB + tid] -= D[(int64_t)j * NB + tid] * (part[0][tid] + part[1][tid]);
}

// backward steps k and k-1 in one launch (the sweep is a chain of nt dependent launches of ~13 us each: two panels per
// launch halve it).  Every workgroup recomputes x_k, the correction of y_{k-1} and x_{k-1} (three 128 x 128 products on
// L2-resident tiles); block 0 stores them, block j+1 applies both panels to y_j, j < k-1.
template <typename T>
__device__ __forceinline__ void bwd_pair_body(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                     const T *__restrict__ D, T *__restrict__ y, T *__restrict__ x, int k,
                                     const int *__restrict__ cols, const int bid) {
  __shared__ T zk[NB], xk[NB], xk1[NB], part[2][NB];
  const int tid = threadIdx.x;
  const int c = tid & (NB - 1), half = tid >> 7;
  auto matvec_t = [&](const T *M, const T *v) {  // part[half][c] = sum over this half's 64 rows of M[r][c] v[r]
    T s = 0;
    for (int r = half * 64; r < half * 64 + 64; r++) s += M[r * NB + c] * v[r];
    part[half][c] = s;
  };
  // the tile indices first (compressed block-sparse storage: a tile outside the pattern does not exist): their table
  // look-ups are dependent global loads, which left where they are used put three memory round trips on the chain of a
  // launch that has nothing to hide them behind (19 -> 32 us per launch on Venice when the tables came in)
  const int j = bid == 0 ? 0 : (cols ? cols[bid - 1] : bid - 1);  // 0 .. k-2
  const int64_t tkk = tix(co, k, k - 1), tkj = tix(co, k, j), tk1j = tix(co, k - 1, j);
  if (tid < NB) zk[tid] = y[(int64_t)k * NB + tid] / D[(int64_t)k * NB + tid];
  __syncthreads();
  matvec_t(Linv + (int64_t)k * NB * NB, zk);
  __syncthreads();
  if (tid < NB) xk[tid] = part[0][tid] + part[1][tid];
  __syncthreads();
  if (tkk >= 0) matvec_t(S + tkk * NB * NB, xk);  // L_{k,k-1}' x_k
  else part[half][c] = 0;
  __syncthreads();
  if (tid < NB) {
    const T d = D[(int64_t)(k - 1) * NB + tid];
    zk[tid] = (y[(int64_t)(k - 1) * NB + tid] - d * (part[0][tid] + part[1][tid])) / d;
  }
  __syncthreads();
  matvec_t(Linv + (int64_t)(k - 1) * NB * NB, zk);
  __syncthreads();
  if (tid < NB) xk1[tid] = part[0][tid] + part[1][tid];
  __syncthreads();
  if (bid == 0) {
    if (tid < NB) {
      x[(int64_t)k * NB + tid] = xk[tid];
      x[(int64_t)(k - 1) * NB + tid] = xk1[tid];
    }
    return;
  }
  // cols (block-sparse S): the tile columns in the pattern of row k or row k-1 (a tile outside the pattern holds zeros)
  const T *Lkj = S + tkj * NB * NB, *Lk1j = S + tk1j * NB * NB;
  T s = 0;
  if (tkj >= 0 && tk1j >= 0) {
    for (int r = half * 64; r < half * 64 + 64; r++) s += Lkj[r * NB + c] * xk[r] + Lk1j[r * NB + c] * xk1[r];
  } else if (tkj >= 0) {
    for (int r = half * 64; r < half * 64 + 64; r++) s += Lkj[r * NB + c] * xk[r];
  } else if (tk1j >= 0) {
    for (int r = half * 64; r < half * 64 + 64; r++) s += Lk1j[r * NB + c] * xk1[r];
  }
  part[half][c] = s;
  __syncthreads();
  if (tid < NB) y[(int64_t)j * NB + tid] -= D[(int64_t)j * NB + tid] * (part[0][tid] + part[1][tid]);
}

template <typename T>
__global__ __launch_bounds__(256) void k_bwd_pair(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                                   const T *__restrict__ D, T *__restrict__ y, T *__restrict__ x, int k,
                                                   const int *__restrict__ cols = nullptr) {
  bwd_pair_body<T>(S, co, Linv, D, y, x, k, cols, (int)blockIdx.x);
}
// The row pairs of TWO independent groups of tile columns in one launch (block-sparse S eliminated from both ends,
// dense_ldl_solve): rows (ka, ka-1) by the first na workgroups, rows (kc, kc-1) by the others.  The two groups' rows have no
// pattern column in common, so the two halves update disjoint parts of y.
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_pair2(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                                    const T *__restrict__ D, T *__restrict__ y, T *__restrict__ x, int ka,
                                                    const int *__restrict__ cols_a, int na, int kc, const int *__restrict__ cols_c) {
  const bool second = (int)blockIdx.x >= na;
  bwd_pair_body<T>(S, co, Linv, D, y, x, second ? kc : ka, second ? cols_c : cols_a, (int)blockIdx.x - (second ? na : 0));
}


}  // namespace

int64_t dense_ldl_tiles_doubles(int64_t n_unpadded) {
  int64_t nt = (n_unpadded + NB - 1) / NB;
  if (nt < 1) nt = 1;
  return nt * (nt + 1) / 2 * NB * NB;
}

template <typename T>
static int set_kernel_attrs() {
  static std::atomic<bool> g_attr_done{false};
  if (g_attr_done.load()) return BA_OK;
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_diag<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAG_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update<T, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_priv_lds_bytes<T>()));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_pairdiag<T, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAG_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_pairdiag<T, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAG_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_pairtrsm<T, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PT_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_pairtrsm<T, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PT_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_rs<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RS_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update2<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)gemm_priv_lds_bytes<T>()));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_diag2<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(DIAG_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_part2<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)part_lds_bytes<T>()));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_part<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds_bytes<T>()));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update<T, 1, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_priv_lds_bytes<T>()));
  g_attr_done.store(true);
  return BA_OK;
}

void dense_ldl_layout(int64_t nt, int world, std::vector<int64_t> *col_off, std::vector<int64_t> *own_range) {
  col_off->assign((size_t)nt, 0);
  if (own_range) own_range->assign((size_t)world + 1, 0);
  int64_t off = 0;
  for (int r = 0; r < world; r++) {
    if (own_range) (*own_range)[(size_t)r] = off;
    for (int64_t q = r; 2 * q < nt; q += world)
      for (int64_t j = 2 * q; j < 2 * q + 2 && j < nt; j++) {
        (*col_off)[(size_t)j] = off;
        off += nt - j;
      }
  }
  if (own_range) (*own_range)[(size_t)world] = off;
}

template <typename T>
int dense_ldl_alloc(DenseLDLT<T> *w, int64_t n_unpadded, T *external_S, int world, int rank, bool lazy_S, bool own_only) {
  int64_t nt = (n_unpadded + NB - 1) / NB;
  if (nt < 1) nt = 1;
  w->n = nt * NB;
  w->nt = nt;
  w->world = world;
  w->rank = rank;
  dense_ldl_layout(nt, world, &w->h_glob_off, &w->own_range);
  w->h_col_tab.assign((size_t)nt + 1, 0);  // head 0: dense layout
  for (int64_t j = 0; j < nt; j++) w->hco()[j] = w->h_glob_off[(size_t)j];
  w->own_only = own_only && world > 1;
  w->s_tiles = nt * (nt + 1) / 2;
  if (w->own_only) {
    // Per-rank ownership: S holds this rank's tile columns only (one contiguous range of the owner-major layout).  The
    // offset table the kernels use becomes rank-local; the columns of other ranks get a negative offset, which no kernel of
    // the distributed factorisation dereferences (the assembly writes through per-chunk tables, ba_lm.hip).
    const int64_t base = w->own_range[(size_t)rank];
    for (int64_t j = 0; j < nt; j++)
      w->hco()[j] = ((j / 2) % world == rank) ? w->hco()[j] - base : BA_NO_TILE;
    w->s_tiles = std::max<int64_t>(1, w->own_range[(size_t)rank + 1] - base);
  }
  BA_HIP_CHECK(hipMalloc((void **)&w->col_tab, w->h_col_tab.size() * sizeof(int64_t)));
  BA_HIP_CHECK(hipMemcpy(w->col_tab, w->h_col_tab.data(), w->h_col_tab.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  w->col_off = w->col_tab + 1;
  // the tile columns this rank owns (pairs q with q % world == rank), ascending, with running tile counts
  w->h_own_cols.clear();
  w->h_own_pref.assign(1, 0);
  for (int64_t j = 0; j < nt; j++)
    if ((j / 2) % world == rank) {
      w->h_own_cols.push_back((int)j);
      w->h_own_pref.push_back(w->h_own_pref.back() + (nt - j));
    }
  BA_HIP_CHECK(hipMalloc((void **)&w->own_cols, (w->h_own_cols.size() + 1) * sizeof(int)));
  BA_HIP_CHECK(hipMalloc((void **)&w->own_pref, w->h_own_pref.size() * sizeof(int64_t)));
  if (!w->h_own_cols.empty())
    BA_HIP_CHECK(hipMemcpy(w->own_cols, w->h_own_cols.data(), w->h_own_cols.size() * sizeof(int), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipMemcpy(w->own_pref, w->h_own_pref.data(), w->h_own_pref.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipMalloc((void **)&w->flag_sum, sizeof(double)));
  if (external_S) {
    w->S = external_S;
    w->own_S = false;
  } else if (!lazy_S) {
    BA_HIP_CHECK(hipMalloc((void **)&w->S, (size_t)w->s_tiles * NB * NB * sizeof(T)));
    w->own_S = true;
  }
  if (!lazy_S) BA_HIP_CHECK(hipMalloc((void **)&w->V, (size_t)8 * nt * NB * NB * sizeof(T)));  // (two runs) x (two slots) x two panels of L*D
  if (!lazy_S && w->own_only) {
    BA_HIP_CHECK(hipMalloc((void **)&w->Lb, (size_t)4 * nt * NB * NB * sizeof(T)));  // their L = V D^-1
    BA_HIP_CHECK(hipMalloc((void **)&w->bpart, (size_t)2 * nt * NB * sizeof(T)));
  }
  BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_chain, hipEventDisableTiming | hipEventReleaseToDevice));
  BA_HIP_CHECK(hipMalloc((void **)&w->Linv, (size_t)nt * NB * NB * sizeof(T)));
  // k_ldl_diag writes the lower triangle of each inverse only; the consumers read whole tiles
  BA_HIP_CHECK(hipMemset(w->Linv, 0, (size_t)nt * NB * NB * sizeof(T)));
  BA_HIP_CHECK(hipDeviceSynchronize());  // (a null-stream memset is not ordered against the non-blocking streams that use Linv)
  BA_HIP_CHECK(hipMalloc((void **)&w->D, (size_t)nt * NB * 2 * sizeof(T)));  // D | y scratch
  BA_HIP_CHECK(hipMalloc((void **)&w->flag, sizeof(int)));
  BA_HIP_CHECK(hipMalloc((void **)&w->ready, (size_t)nt * sizeof(int)));
  BA_HIP_CHECK(hipStreamCreateWithFlags(&w->hoist, hipStreamNonBlocking));
  {  // the stream of the block-sparse schedule's look-ahead: lowest priority, so that the panel chain's kernels are placed first
    int lo = 0, hi = 0;
    BA_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    BA_HIP_CHECK(hipStreamCreateWithPriority(&w->rest, hipStreamNonBlocking, lo));
  }
  BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_top, hipEventDisableTiming | hipEventReleaseToDevice));
  // the distributed factorisation's events order work whose data LEAVES the device (panels handed to the transport):
  // default release scope, not hipEventReleaseToDevice as the single-GPU hoist events above
  BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_dtop, hipEventDisableTiming));
  BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_dchain, hipEventDisableTiming));
  for (int q = 0; q < 2; q++) {
    BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_recv[q], hipEventDisableTiming));
    BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_upd[q], hipEventDisableTiming));
  }
  return set_kernel_attrs<T>();
}

template <typename T>
int dense_ldl_alloc_S(DenseLDLT<T> *w) {
  if (!w->S) {
    BA_HIP_CHECK(hipMalloc((void **)&w->S, (size_t)w->s_tiles * NB * NB * sizeof(T)));
    w->own_S = true;
  }
  if (!w->V) BA_HIP_CHECK(hipMalloc((void **)&w->V, (size_t)8 * w->nt * NB * NB * sizeof(T)));
  if (!w->Lb && w->own_only) BA_HIP_CHECK(hipMalloc((void **)&w->Lb, (size_t)4 * w->nt * NB * NB * sizeof(T)));
  if (!w->bpart && w->own_only) BA_HIP_CHECK(hipMalloc((void **)&w->bpart, (size_t)2 * w->nt * NB * sizeof(T)));
  return BA_OK;
}

template <typename T>
void dense_ldl_free(DenseLDLT<T> *w) {
  if (w->own_S && w->S) (void)hipFree(w->S);
  if (w->V) (void)hipFree(w->V);
  if (w->Lb) (void)hipFree(w->Lb);
  if (w->bpart) (void)hipFree(w->bpart);
  if (w->Linv) (void)hipFree(w->Linv);
  if (w->D) (void)hipFree(w->D);
  if (w->flag) (void)hipFree(w->flag);
  if (w->col_tab) (void)hipFree(w->col_tab);
  if (w->own_cols) (void)hipFree(w->own_cols);
  if (w->own_pref) (void)hipFree(w->own_pref);
  if (w->flag_sum) (void)hipFree(w->flag_sum);
  if (w->ready) (void)hipFree(w->ready);
  if (w->prow) (void)hipFree(w->prow);
  if (w->lcol) (void)hipFree(w->lcol);
  if (w->lpair) (void)hipFree(w->lpair);
  if (w->upd_ij) (void)hipFree(w->upd_ij);
  if (w->own_tiles) (void)hipFree(w->own_tiles);
  if (w->hoist) (void)hipStreamDestroy(w->hoist);
  if (w->rest) (void)hipStreamDestroy(w->rest);
  if (w->ev_top) (void)hipEventDestroy(w->ev_top);
  if (w->ev_chain) (void)hipEventDestroy(w->ev_chain);
  if (w->ev_dtop) (void)hipEventDestroy(w->ev_dtop);
  if (w->ev_dchain) (void)hipEventDestroy(w->ev_dchain);
  for (int q = 0; q < 2; q++) {
    if (w->ev_recv[q]) (void)hipEventDestroy(w->ev_recv[q]);
    if (w->ev_upd[q]) (void)hipEventDestroy(w->ev_upd[q]);
  }
  *w = DenseLDLT<T>();
}

template <typename T>
static int launch_diag(ba_problem *p, DenseLDLT<T> *w, int k, hipStream_t st, const int *wait_ready = nullptr, bool clear_flag = false) {
  ProfScope ps(p, PC_LDL_DIAG, st);
  hipLaunchKernelGGL(k_ldl_diag<T>, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_ELEMS * sizeof(T), st, w->S + tix(w->hco(), k, k) * NB * NB,
                     w->Linv + (int64_t)k * NB * NB, w->D + (int64_t)k * NB, w->flag, (unsigned long long *)nullptr,
                     wait_ready, clear_flag ? 1 : 0);
  return BA_OK;
}

// b != null: forward substitution of b fused (y_k and b_i -= L_ik y_k); the last panel has no tile below it, its y_k
// comes from the stand-alone forward step kernel.
template <typename T>
static int launch_trsm(ba_problem *p, DenseLDLT<T> *w, int k, T *V, T *b, hipStream_t st) {
  const int m = (int)w->nt - k - 1;
  T *y = w->D + w->nt * NB;
  if (m <= 0) {
    if (b) hipLaunchKernelGGL(k_fwd_step<T>, dim3(1), dim3(256), 0, st, w->S, w->col_off, w->Linv, b, y, k);
    return BA_OK;
  }
  ProfScope ps(p, PC_LDL_TRSM, st);
  if (b)
    hipLaunchKernelGGL((k_ldl_trsm_rs<T, true>), dim3(4 * m), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, w->Linv + (int64_t)k * NB * NB,
                       w->D + (int64_t)k * NB, V, k, b, y);
  else
    hipLaunchKernelGGL((k_ldl_trsm_rs<T, false>), dim3(4 * m), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, w->Linv + (int64_t)k * NB * NB,
                       w->D + (int64_t)k * NB, V, k, b, y);
  return BA_OK;
}

template <typename T>
static int launch_col(ba_problem *p, DenseLDLT<T> *w, int k, const T *V0, hipStream_t st) {
  const int m = (int)w->nt - k - 1;
  if (m <= 0) return BA_OK;
  ProfScope ps(p, PC_LDL_SYRK, st);
  hipLaunchKernelGGL(k_ldl_col_rs<T>, dim3(4 * m), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, V0, k);
  return BA_OK;
}

// pair update of the lower tiles (i, j), base <= j <= i, with panels k, k+1; `ready`: flag raised when tile (base, base)
// is final (hoisted-diagonal schedule)
// tiles up to which the pair update takes its row-split form (BA_LDL_UPDATE_RS_MAX; 0 disables)
static int update_rs_max() {
  const char *e = getenv("BA_LDL_UPDATE_RS_MAX");  // read per call: a test compares the two kernels in one process
  return e ? atoi(e) : 320;
}

template <typename T>
static int launch_pair(ba_problem *p, DenseLDLT<T> *w, int k, int base, const T *V0, const T *V1, hipStream_t st,
                       int *ready = nullptr, int ready_tiles = 1) {
  const int nt = (int)w->nt, m = nt - base;
  if (m <= 0) return BA_OK;
  const int nblk = m * (m + 1) / 2;
  if (!ready && nblk <= update_rs_max()) {  // short update: row-split form (see k_ldl_update_rs)
    ProfScope ps(p, PC_LDL_UPDATE_RS, st);
    hipLaunchKernelGGL(k_ldl_update_rs<T>, dim3(4 * nblk), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, V0, V1, k, base, nblk,
                       (const int *)nullptr);
    return BA_OK;
  }
  ProfScope ps(p, PC_LDL_UPDATE, st);
  // (Round 3, tried and removed: the remainder of a launch beyond its full rounds of 512 tiles sent ahead in row-split form
  // (k_ldl_update_rs) so that the big kernel runs full rounds only -- Venice 40.2-40.5 -> 41.3-41.5 ms per LM iteration: per
  // tile the row-split kernel reads its B tiles four times and costs more than the partly filled round it replaces.  The
  // same slices as the LAST BLOCKS of the big kernel itself put the two bodies' registers together: 288 VGPRs, one
  // workgroup per CU.)
  // (Cutting the tiles of a partly filled last round into 64 x 64 quadrants, one workgroup each, was tried and removed:
  // 34.1-34.3 ms against 33.9-34.1 at n = 16 002.  A partial round does not cost a full one -- the quadrant kernel took
  // 39 us on average, which is what the big kernel's own last round costs.)
  static const int blocked = [] { const char *e = getenv("BA_LDL_TRI_BLOCKED"); return e ? atoi(e) : TSB; }();
  hipLaunchKernelGGL((k_ldl_update<T, 1>), dim3(((nblk + 7) / 8) * 8), dim3(256), gemm_priv_lds_bytes<T>(), st, w->S,
                     w->col_off, V0, V1, k, base, nt, nblk, ready, (const int *)nullptr, (const int64_t *)nullptr, 0, 0,
                     ready_tiles, (const int *)nullptr, (const T *)nullptr, (const T *)nullptr, (const int2 *)nullptr, blocked);
  return BA_OK;
}

// the fused panel-pair kernels (see k_ldl_pairdiag): b != null: forward substitution rides along
template <typename T>
static int launch_pairdiag(ba_problem *p, DenseLDLT<T> *w, int k, T *V0, T *b, hipStream_t st, const int *wait_ready, int need) {
  ProfScope ps(p, PC_LDL_DIAG, st);
  T *y = w->D + w->nt * NB;
  if (b)
    hipLaunchKernelGGL((k_ldl_pairdiag<T, true>), dim3(1), dim3(256), DIAG_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, w->Linv, w->D,
                       V0, k, (int)w->nt, w->flag, wait_ready, need, b, y);
  else
    hipLaunchKernelGGL((k_ldl_pairdiag<T, false>), dim3(1), dim3(256), DIAG_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, w->Linv, w->D,
                       V0, k, (int)w->nt, w->flag, wait_ready, need, b, y);
  return BA_OK;
}

template <typename T>
static int launch_pairtrsm(ba_problem *p, DenseLDLT<T> *w, int k, T *V0, T *V1, T *b, hipStream_t st) {
  const int m = (int)w->nt - k - 2;
  if (m <= 0) return BA_OK;
  ProfScope ps(p, PC_LDL_TRSM, st);
  const T *y = w->D + w->nt * NB;
  if (b)
    hipLaunchKernelGGL((k_ldl_pairtrsm<T, true>), dim3(4 * m), dim3(256), PT_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, w->Linv,
                       w->D, V0, V1, k, b, y);
  else
    hipLaunchKernelGGL((k_ldl_pairtrsm<T, false>), dim3(4 * m), dim3(256), PT_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, w->Linv,
                       w->D, V0, V1, k, b, y);
  return BA_OK;
}

// Two panels per pass over the trailing matrix:
//   diag(k) trsm(k) | column update of tile column k+1 | diag(k+1) trsm(k+1) | pair update of everything right of k+1.
//
// Hoisted diagonal tile (for nt >= HOIST_MIN_TILES + 2; BA_LDL_HOIST=0 disables; never with per-kernel profiling, whose
// event pairs must time single kernels): the kernel that factors tile (k+2, k+2) is LAUNCHED ahead of time on a second
// stream -- while CUs are free; once the trailing update fills the GPU a 153 KB-LDS workgroup finds no CU -- and waits in
// place for the flag the trailing update's first workgroup raises when that tile is final.  It then factors the tile
// beside the rest of the update: one 59 us diagonal kernel per pair leaves the critical path (37.2 -> 35.4 ms at
// n = 16002).  Nothing throughput-bound is moved and no CU mask is involved.  (A full look-ahead -- whole panel chain on a
// priority stream, bulk on a CU-masked stream -- measured 40-51 ms; masking only 1 / 2 / 4 of the 256 CUs off the bulk
// stream for a hoisted diagonal kernel 42 / 46 / 59 ms: CU-masked streams are slow here, and that code is gone.)
template <typename T>
int dense_ldl_factor_sparse(ba_problem *p, DenseLDLT<T> *w, hipStream_t st, int *zero_pivot, T *d_b);

template <typename T>
int dense_ldl_factor(ba_problem *p, DenseLDLT<T> *w, hipStream_t st, int *zero_pivot, T *d_b) {
  if (w->sparse && !p->comm.active()) return dense_ldl_factor_sparse(p, w, st, zero_pivot, d_b);
  const int nt = (int)w->nt;
  const int64_t panel = (int64_t)nt * NB * NB;
  T *Vs[2][2] = {{w->V, w->V + panel}, {w->V + 2 * panel, w->V + 3 * panel}};
  constexpr int HOIST_MIN_TILES = 32;  // below ~2 rounds of tiles the update is shorter than wait + factor
  // The waiting workgroup keeps one CU of one XCD from the update, whose blocks the hardware deals round-robin to the
  // XCDs: that XCD runs 32/31 longer and the launch ends with it -- 3 % of the update time, which grows as nt^3 while
  // the hoisted 59 us per pair grow as nt.  Measured: n = 16002 (nt 126) 37.1 -> 35.6 ms, n = 40000 (nt 313) 411 -> 418 ms;
  // the model's break-even is nt ~ 250.
  constexpr int HOIST_MAX_TILES = 224;
  static const bool hoist_off = [] { const char *e = getenv("BA_LDL_HOIST"); return e && e[0] == '0'; }();
  w->hoisting = !p->prof_on && !hoist_off && !w->hoist_disabled && !p->comm.active() && nt >= HOIST_MIN_TILES + 2 && nt <= HOIST_MAX_TILES;
  // the pivot flag: hoisted kernels read it (an earlier tile gave up) before tile 0 is factored -- cleared ahead of the
  // fork; in order, the first diagonal kernel clears it
  if (w->hoisting) {
    BA_HIP_CHECK(hipMemsetAsync(w->flag, 0, sizeof(int), st));
    BA_HIP_CHECK(hipMemsetAsync(w->ready, 0, (size_t)nt * sizeof(int), st));
    // one fork for the whole factorisation: the hoisted kernels only depend on their flags (and on stream order among
    // themselves); each gets its CU in the idle gaps of the panel chain before the trailing update it waits for starts
    BA_HIP_CHECK(hipEventRecord(w->ev_top, st));
    BA_HIP_CHECK(hipStreamWaitEvent(w->hoist, w->ev_top, 0));
  }
  // Fused pair schedule (see k_ldl_pairdiag): pair (k, k+1) is "fused" when its three leading tiles were factored by ONE
  // hoisted workgroup beside the previous pair's trailing update; what then remains between two trailing updates is a single
  // row-parallel kernel.  It pays while that update is long enough to hide the ~190 us of the hoisted workgroup (two
  // diagonal tiles and two 128^3 products in sequence): FUSE_MIN_TILES tile rows.  Shorter updates keep round 1's schedule
  // (one hoisted diagonal tile down to HOIST_MIN_TILES, strictly in order below that).  BA_LDL_FUSE=0 disables.
  static const int fuse_min = [] { const char *e = getenv("BA_LDL_FUSE_MIN"); return e ? atoi(e) : 48; }();
  static const bool fuse_off = [] { const char *e = getenv("BA_LDL_FUSE"); return e && e[0] == '0'; }();
  auto fused = [&](int k) { return w->hoisting && !fuse_off && k >= 2 && k + 1 < nt && nt - k >= fuse_min; };
  launch_diag(p, w, 0, st, nullptr, !w->hoisting);
  for (int k = 0, q = 0; k < nt; k += 2, q ^= 1) {
    T *V0 = Vs[q][0], *V1 = Vs[q][1];
    const bool more = k + 2 < nt;
    const bool next_fused = more && fused(k + 2);
    const bool next_hoist1 = more && !next_fused && w->hoisting && (nt - k - 2 >= HOIST_MIN_TILES);
    const int need = (k + 3 < nt) ? 3 : 1;
    if (next_fused) {  // the next pair's leading tiles: waits in place for `need` tiles of this pair's trailing update
      launch_pairdiag(p, w, k + 2, Vs[q ^ 1][0], d_b, w->hoist, w->ready + k + 2, need);
      BA_HIP_CHECK(hipEventRecord(w->ev_chain, w->hoist));
    } else if (next_hoist1) {  // the next pair's first diagonal tile only
      launch_diag(p, w, k + 2, w->hoist, w->ready + k + 2);
      BA_HIP_CHECK(hipEventRecord(w->ev_chain, w->hoist));
    }
    if (fused(k)) {
      launch_pairtrsm(p, w, k, V0, V1, d_b, st);
    } else {
      launch_trsm(p, w, k, V0, d_b, st);  // diag(k) is done: first tile, hoisted, or the in-order branch below
      if (k + 1 < nt) {
        launch_col(p, w, k, V0, st);
        launch_diag(p, w, k + 1, st);
        launch_trsm(p, w, k + 1, V1, d_b, st);
      }
    }
    if (!more) break;
    const bool hoisted = next_fused || next_hoist1;
    launch_pair(p, w, k, k + 2, V0, V1, st, hoisted ? w->ready + k + 2 : nullptr, next_fused ? need : 1);
    if (hoisted)
      BA_HIP_CHECK(hipStreamWaitEvent(st, w->ev_chain, 0));  // join: the hoisted workgroup's outputs are written
    else
      launch_diag(p, w, k + 2, st);
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

// ---- block-sparse reduced camera system ---------------------------------------------------------------------------------
// Real bundle-adjustment problems do not connect every camera pair: S has empty 9 x 9 blocks, and the reference's sparse
// LDL' exploits that (symbolic phase src/ldl_aux.jl:82-119, numeric src/ldl_aux.jl:122-201).  Here the sparsity is kept at
// the granularity the matrix cores work in: a tile is in the pattern when some camera pair of the Schur key list lands in
// it, and the symbolic factorisation is done per tile column PAIR (the unit of the schedule): the rows U_q of pair q are
// the tile rows below it with a pattern tile in either column, and every tile (i, j), i >= j, i, j in U_q, joins the
// pattern (fill).  Taking the union of the two columns' rows keeps ONE row list per pair; a tile whose operands are
// structurally zero receives a zero update (correct, a little wasted work when the two columns differ).
// The numeric phase is the in-order pair schedule with row lists: diag(k), panel solve of column k over {k+1} + U_q,
// column update, diag(k+1), panel solve of column k+1 over U_q, pair update over the lower tiles of U_q x U_q.  Tiles
// outside the pattern are never read or written: they hold the zeros of the assembly's memset.
// (the symbolic phase, tile_pattern_build, is host-only code: ba_order.cpp)

template <typename T>
int dense_ldl_use_pattern(DenseLDLT<T> *w, const TilePattern *pat) {
  for (void *q : {(void *)w->prow, (void *)w->lcol, (void *)w->lpair, (void *)w->upd_ij, (void *)w->own_tiles})
    if (q) (void)hipFree(q);
  w->prow = w->lcol = w->lpair = nullptr;
  w->upd_ij = w->own_tiles = nullptr;
  w->pat = pat;
  w->sparse = pat != nullptr;
  if (!pat) return BA_OK;
  {
    // Compressed storage: only the tiles of the pattern are allocated.  Column j (pair q = j / 2, k = 2q) stores the tile rows
    // j, k+1 (even j), U_q, in that order; the table behind the column offsets gives every (i, j) its position (tix).
    // On several ranks (per-rank ownership) a rank stores the columns of ITS pairs only, one contiguous range of the
    // owner-major order; the other columns get a negative offset.
    const int64_t nt = w->nt;
    const int P = w->own_only ? w->world : 1, me = w->own_only ? w->rank : 0;
    std::vector<int64_t> tab((size_t)(1 + nt + nt * nt), BA_NO_TILE);
    tab[0] = nt;
    int64_t *co = tab.data() + 1, *rp = tab.data() + 1 + nt;
    w->h_col_cnt.assign((size_t)nt, 0);
    for (int64_t j = 0; j < nt; j++) {
      const int64_t q = j / 2;
      int64_t pos = 0;
      rp[j * nt + j] = pos++;
      const int l0 = pat->prow_ptr[(size_t)q], l1 = pat->prow_ptr[(size_t)q + 1];
      for (int l = l0; l < l1; l++) {
        const int64_t i = pat->prow[(size_t)l];  // [k+1] + U_q
        if (i > j) rp[i * nt + j] = pos++;
      }
      w->h_col_cnt[(size_t)j] = pos;
    }
    w->own_range.assign((size_t)P + 1, 0);
    int64_t run = 0;
    for (int r = 0; r < P; r++) {
      w->own_range[(size_t)r] = run;
      for (int64_t j = 0; j < nt; j++)
        if ((j / 2) % P == r) {
          if (r == me) co[j] = run;  // (made rank-local below)
          run += w->h_col_cnt[(size_t)j];
        }
    }
    w->own_range[(size_t)P] = run;
    const int64_t base = w->own_range[(size_t)me];
    for (int64_t j = 0; j < nt; j++)
      if (co[j] >= 0) co[j] -= base;
    w->s_tiles = std::max<int64_t>(1, w->own_range[(size_t)me + 1] - base);
    if (w->own_only) {
      // the running tile counts of the owned columns (k_ldl_update's OWN mode is not used with a pattern; the column scaling
      // and the chunked assembly are), the stored tiles in storage order, and the update lists per pair
      w->h_own_pref.assign(1, 0);
      std::vector<int2> tiles;
      for (int j : w->h_own_cols) {
        w->h_own_pref.push_back(w->h_own_pref.back() + w->h_col_cnt[(size_t)j]);
        const int64_t q = j / 2;
        tiles.push_back(make_int2(j, j));
        for (int l = pat->prow_ptr[(size_t)q]; l < pat->prow_ptr[(size_t)q + 1]; l++)
          if (pat->prow[(size_t)l] > j) tiles.push_back(make_int2(pat->prow[(size_t)l], j));
      }
      BA_HIP_CHECK(hipMemcpy(w->own_pref, w->h_own_pref.data(), w->h_own_pref.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      BA_HIP_CHECK(hipMalloc((void **)&w->own_tiles, (tiles.size() + 1) * sizeof(int2)));
      if (!tiles.empty()) BA_HIP_CHECK(hipMemcpy(w->own_tiles, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
      std::vector<int2> upd;
      const int npairs = (int)((nt + 1) / 2);
      w->h_upd_ptr.assign(1, 0);
      w->h_upd_lead.assign((size_t)npairs, 0);
      for (int q = 0; q < npairs; q++) {
        const int k = 2 * q, l0 = pat->prow_ptr[(size_t)q], l1 = pat->prow_ptr[(size_t)q + 1];
        for (int a = l0 + 1; a < l1; a++) {  // U_q (the list starts with tile row k+1)
          const int j = pat->prow[(size_t)a];
          if ((j / 2) % P != me) continue;
          for (int c = a; c < l1; c++) upd.push_back(make_int2(pat->prow[(size_t)c], j));
          if (j < k + 4) w->h_upd_lead[(size_t)q] = (int)upd.size() - w->h_upd_ptr.back();
        }
        w->h_upd_ptr.push_back((int)upd.size());
      }
      BA_HIP_CHECK(hipMalloc((void **)&w->upd_ij, (upd.size() + 1) * sizeof(int2)));
      if (!upd.empty()) BA_HIP_CHECK(hipMemcpy(w->upd_ij, upd.data(), upd.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    w->h_col_tab.swap(tab);
    if (w->col_tab) (void)hipFree(w->col_tab);
    w->col_tab = nullptr;
    BA_HIP_CHECK(hipMalloc((void **)&w->col_tab, w->h_col_tab.size() * sizeof(int64_t)));
    BA_HIP_CHECK(hipMemcpy(w->col_tab, w->h_col_tab.data(), w->h_col_tab.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    w->col_off = w->col_tab + 1;
    if (w->S && w->own_S) {  // (a workspace allocated before its pattern was known: Float32 twin)
      (void)hipFree(w->S);
      w->S = nullptr;
      BA_HIP_CHECK(hipMalloc((void **)&w->S, (size_t)w->s_tiles * NB * NB * sizeof(T)));
    }
  }
  BA_HIP_CHECK(hipMalloc((void **)&w->prow, (pat->prow.size() + 1) * sizeof(int)));
  BA_HIP_CHECK(hipMalloc((void **)&w->lcol, (pat->lcol.size() + 1) * sizeof(int)));
  if (!pat->prow.empty()) BA_HIP_CHECK(hipMemcpy(w->prow, pat->prow.data(), pat->prow.size() * sizeof(int), hipMemcpyHostToDevice));
  if (!pat->lcol.empty()) BA_HIP_CHECK(hipMemcpy(w->lcol, pat->lcol.data(), pat->lcol.size() * sizeof(int), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipMalloc((void **)&w->lpair, (pat->lpair.size() + 1) * sizeof(int)));
  if (!pat->lpair.empty()) BA_HIP_CHECK(hipMemcpy(w->lpair, pat->lpair.data(), pat->lpair.size() * sizeof(int), hipMemcpyHostToDevice));
  return BA_OK;
}

// The pair schedule over the pattern (see above); d_b != null: the forward substitution rides along.
//
// Look-ahead of one pair (BA_SPARSE_LOOKAHEAD=0 disables; never with per-kernel profiling): of pair q's trailing update the
// next pair's panel chain needs only the tiles of tile columns k+2, k+3 -- the "lead" part, at most two columns of the row
// list, done at once on the main stream -- while the rest (the lower triangle over the remaining rows) runs on a second
// stream beside that chain: diag(k+2), panel solve, column update, diag(k+3), panel solve are one workgroup or a few dozen
// each and leave the chip almost empty.  The rest keeps to a part of the chip (k_ldl_update_part: the diagonal-tile kernel
// needs a whole CU).  Ordering: lead(q) waits for rest(q-1) (both update tiles of columns k+2, k+3), rest(q) starts behind
// lead(q) (beside it the lead took as long as the whole update), rests follow one another on their stream, and the panel
// buffers alternate between pairs (rest(q) reads the buffers pair q+2 writes: it has been joined by then).  The parts touch
// disjoint tiles and each tile receives its updates in the same order as in the in-order schedule: same bits (tested).
// Forks and joins are events, so the factorisation still records into one hipGraph.
// What it buys (kernel trace, Final-13682 shape with 6 % tile fill, Float32): the in-order pair takes 137 us -- diag 28,
// solve 11, column 8, diag 28, solve 11, update 53; with the look-ahead the 48 us rest disappears behind the next chain, the
// 12 us lead stays, and every fork / join of the replayed graph costs ~12 us of cross-queue synchronisation: 131 us per pair,
// 101 -> 93 ms per LM iteration.  Rests shorter than BA_SPARSE_LOOKAHEAD_MIN tiles stay in order: 96 (sweep on the Venice shape
// with locality 0.13, whose rests are 120 tiles, two runs per launch: 16 / 48 / 96 / 160 / 256 -> 11.8 / 11.8 / 11.8 / 12.7 /
// 12.7 ms; the Final shape is indifferent).  What shortens the chain itself: two independent runs of pairs advancing in
// the same launches, below ("Two runs").
template <typename T>
int dense_ldl_factor_sparse(ba_problem *p, DenseLDLT<T> *w, hipStream_t st, int *zero_pivot, T *d_b) {
  const int nt = (int)w->nt;
  const TilePattern *pat = w->pat;
  const int64_t panel = (int64_t)nt * NB * NB;
  const int npairs = (nt + 1) / 2;
  T *y = w->D + w->nt * NB;
  w->hoisting = false;
  const char *la_env = getenv("BA_SPARSE_LOOKAHEAD");  // read per call: tests compare both schedules in one process
  const bool lookahead = !p->prof_on && !(la_env && la_env[0] == '0');
  const int la_min = [] { const char *e = getenv("BA_SPARSE_LOOKAHEAD_MIN"); return e ? atoi(e) : 96; }();  // (read per call: the tests force it to 1)
  static const int rest_cus = [] { const char *e = getenv("BA_SPARSE_REST_CUS"); return e ? atoi(e) : 224; }();  // CUs the rest may take (sweep: 128 / 192 / 224 -> 76.1 / 71.9 / 71.3 ms on the Final shape)
  bool pending[2] = {false, false};  // rest of pair q (slot q & 1) launched on the second stream and not yet joined
  auto join_rest = [&](int slot) -> int {
    if (pending[slot]) {
      BA_HIP_CHECK(hipStreamWaitEvent(st, w->ev_upd[slot], 0));
      pending[slot] = false;
    }
    return BA_OK;
  };
  auto launch_update = [&](hipStream_t s2, const T *V0, const T *V1, int k, const int *rows, int cnt) {
    const int nblk = cnt * (cnt + 1) / 2;
    if (nblk <= 0) return;
    ProfScope ps(p, nblk <= update_rs_max() ? PC_LDL_UPDATE_RS : PC_LDL_UPDATE, s2);
    if (nblk <= update_rs_max())
      hipLaunchKernelGGL(k_ldl_update_rs<T>, dim3(4 * nblk), dim3(256), RS_LDS_ELEMS * sizeof(T), s2, w->S, w->col_off, V0, V1, k, k + 2,
                         nblk, rows, 0);
    else
      hipLaunchKernelGGL((k_ldl_update<T, 1>), dim3(((nblk + 7) / 8) * 8), dim3(256), gemm_priv_lds_bytes<T>(), s2, w->S,
                         w->col_off, V0, V1, k, k + 2, nt, nblk, (int *)nullptr, (const int *)nullptr, (const int64_t *)nullptr, 0, 0,
                         1, rows);
  };
  // the panel chain of pair q on stream s: diag(k), panel solve over {k+1} + U_q, column update, diag(k+1), panel solve over U_q
  // (+ the forward substitution of d_b).  Returns the number of rows of U_q whose trailing update is still to do (0: none).
  auto chain = [&](int q, hipStream_t s, T *V0, T *V1, bool clear_flag) -> int {
    const int k = 2 * q;
    const int l0 = pat->prow_ptr[(size_t)q], l1 = pat->prow_ptr[(size_t)q + 1];
    const int c1 = l1 - l0;                      // {k+1} + U_q
    const int c2 = c1 > 0 ? c1 - 1 : 0;          // U_q
    const int *rows1 = w->prow + l0, *rows2 = w->prow + l0 + 1;
    launch_diag(p, w, k, s, nullptr, clear_flag);
    if (c1 == 0) {  // last, single tile column: y_k only
      if (d_b) hipLaunchKernelGGL(k_fwd_step<T>, dim3(1), dim3(256), 0, s, w->S, w->col_off, w->Linv, d_b, y, k, (const int *)nullptr);
      return 0;
    }
    {
      ProfScope ps(p, PC_LDL_TRSM, s);
      if (d_b)
        hipLaunchKernelGGL((k_ldl_trsm_rs<T, true>), dim3(4 * c1), dim3(256), RS_LDS_ELEMS * sizeof(T), s, w->S, w->col_off,
                           w->Linv + (int64_t)k * NB * NB, w->D + (int64_t)k * NB, V0, k, d_b, y, rows1);
      else
        hipLaunchKernelGGL((k_ldl_trsm_rs<T, false>), dim3(4 * c1), dim3(256), RS_LDS_ELEMS * sizeof(T), s, w->S, w->col_off,
                           w->Linv + (int64_t)k * NB * NB, w->D + (int64_t)k * NB, V0, k, d_b, y, rows1);
    }
    {
      ProfScope ps(p, PC_LDL_SYRK, s);
      hipLaunchKernelGGL(k_ldl_col_rs<T>, dim3(4 * c1), dim3(256), RS_LDS_ELEMS * sizeof(T), s, w->S, w->col_off, V0, k, rows1);
    }
    launch_diag(p, w, k + 1, s);
    if (c2 == 0) {
      if (d_b) hipLaunchKernelGGL(k_fwd_step<T>, dim3(1), dim3(256), 0, s, w->S, w->col_off, w->Linv, d_b, y, k + 1, (const int *)nullptr);
      return 0;
    }
    {
      ProfScope ps(p, PC_LDL_TRSM, s);
      if (d_b)
        hipLaunchKernelGGL((k_ldl_trsm_rs<T, true>), dim3(4 * c2), dim3(256), RS_LDS_ELEMS * sizeof(T), s, w->S, w->col_off,
                           w->Linv + (int64_t)(k + 1) * NB * NB, w->D + (int64_t)(k + 1) * NB, V1, k + 1, d_b, y, rows2);
      else
        hipLaunchKernelGGL((k_ldl_trsm_rs<T, false>), dim3(4 * c2), dim3(256), RS_LDS_ELEMS * sizeof(T), s, w->S, w->col_off,
                           w->Linv + (int64_t)(k + 1) * NB * NB, w->D + (int64_t)(k + 1) * NB, V1, k + 1, d_b, y, rows2);
    }
    return c2;
  };
  // Two runs (TilePattern::a_clean / b_clean: the ordering eliminated a profile from both ends, ba_order.cpp): the first
  // a_clean pairs and the b_clean pairs from `split` touch disjoint tiles, rows of the right-hand side and panel buffers
  // (slot 0 / slot 1).  They advance together, pair i of either run in the SAME launches ("two runs per launch" above): one
  // stream, no event, half the launches -- the kernels of a panel chain are a workgroup or a few dozen each and the latency of
  // a launch with two panels' worth of them is that of one.  The longer run finishes alone; every other pair follows below in
  // index order.  Each tile still receives its updates in ascending pair order WITHIN a run; the frontier's tiles receive the
  // second run's updates before the late pairs of the first -- a different but fixed summation order
  // (BA_SPARSE_TWO_RUNS=0: one chain; never with per-kernel profiling, whose classes time single-run launches).
  const char *tc_env = getenv("BA_SPARSE_TWO_RUNS");
  const bool two = pat->a_clean > 0 && pat->b_clean > 0 && !p->prof_on && !(tc_env && tc_env[0] == '0');
  std::vector<int> order;
  if (two) {
    BA_HIP_CHECK(hipMemsetAsync(w->flag, 0, sizeof(int), st));  // (no diagonal kernel of the two runs clears the pivot flag)
    const int both = std::min(pat->a_clean, pat->b_clean);
    for (int i = 0; i < both; i++) {
      const int slot = i & 1;
      // panel buffers: run A takes panels 0..3 (two slots of two), run B panels 4..7; the slots alternate for the look-ahead
      T *VA = w->V + 2 * slot * panel, *VB = w->V + (4 + 2 * slot) * panel;
      const int qa = i, qb = pat->split + i, ka = 2 * qa, kb = 2 * qb;
      const int la = pat->prow_ptr[(size_t)qa], lb = pat->prow_ptr[(size_t)qb];
      const int c1a = pat->prow_ptr[(size_t)qa + 1] - la, c1b = pat->prow_ptr[(size_t)qb + 1] - lb;  // {k+1} + U_q: >= 2 inside a run
      const int c2a = c1a - 1, c2b = c1b - 1;
      auto tile = [&](int k) { return w->S + tix(w->hco(), k, k) * NB * NB; };
      auto panel_of = [&](int k, T *V, const int *rows) { return RunPanel<T>{w->Linv + (int64_t)k * NB * NB, w->D + (int64_t)k * NB, V, k, rows}; };
      BA_CHECK(join_rest(slot));  // (the rests of the step two back read these panel buffers)
      hipLaunchKernelGGL(k_ldl_diag2<T>, dim3(2), dim3(DIAG_THREADS), DIAG_LDS_ELEMS * sizeof(T), st, tile(ka), w->Linv + (int64_t)ka * NB * NB,
                         w->D + (int64_t)ka * NB, tile(kb), w->Linv + (int64_t)kb * NB * NB, w->D + (int64_t)kb * NB, w->flag);
      const RunPanel<T> a0 = panel_of(ka, VA, w->prow + la), b0 = panel_of(kb, VB, w->prow + lb);
      if (d_b)
        hipLaunchKernelGGL((k_ldl_trsm_rs2<T, true>), dim3(4 * (c1a + c1b)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, a0, b0, 4 * c1a, d_b, y);
      else
        hipLaunchKernelGGL((k_ldl_trsm_rs2<T, false>), dim3(4 * (c1a + c1b)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, a0, b0, 4 * c1a, d_b, y);
      hipLaunchKernelGGL(k_ldl_col_rs2<T>, dim3(4 * (c1a + c1b)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, a0, b0, 4 * c1a);
      hipLaunchKernelGGL(k_ldl_diag2<T>, dim3(2), dim3(DIAG_THREADS), DIAG_LDS_ELEMS * sizeof(T), st, tile(ka + 1),
                         w->Linv + (int64_t)(ka + 1) * NB * NB, w->D + (int64_t)(ka + 1) * NB, tile(kb + 1), w->Linv + (int64_t)(kb + 1) * NB * NB,
                         w->D + (int64_t)(kb + 1) * NB, w->flag);
      const RunPanel<T> a1 = panel_of(ka + 1, VA + panel, w->prow + la + 1), b1 = panel_of(kb + 1, VB + panel, w->prow + lb + 1);
      if (d_b)
        hipLaunchKernelGGL((k_ldl_trsm_rs2<T, true>), dim3(4 * (c2a + c2b)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, a1, b1, 4 * c2a, d_b, y);
      else
        hipLaunchKernelGGL((k_ldl_trsm_rs2<T, false>), dim3(4 * (c2a + c2b)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, a1, b1, 4 * c2a, d_b, y);
      // the two pair updates: panels (k, k+1) of either run over its U_q -- with the look-ahead of the single run (lead strips of
      // both runs first, both rests on a part of the chip beside the next step's chain) when the rests are long enough
      auto nlead_of = [&](int l0, int c2, int k) {
        int nl = 0;
        while (nl < c2 && nl < 2 && pat->prow[(size_t)(l0 + 1 + nl)] < k + 4) nl++;
        return nl;
      };
      const int nla = nlead_of(la, c2a, ka), nlb = nlead_of(lb, c2b, kb);
      const int ra = c2a - nla, rb = c2b - nlb, rest_a = ra * (ra + 1) / 2, rest_b = rb * (rb + 1) / 2;
      if (lookahead && i + 1 < both && rest_a + rest_b >= 2 * la_min) {
        BA_CHECK(join_rest(slot ^ 1));
        const int sa = nla == 0 ? 0 : (nla == 1 ? c2a : 2 * c2a - 1), sb = nlb == 0 ? 0 : (nlb == 1 ? c2b : 2 * c2b - 1);
        const RunPanel<T> ua = panel_of(ka, VA, w->prow + la + 1), ub = panel_of(kb, VB, w->prow + lb + 1);
        if (sa + sb > 0)
          hipLaunchKernelGGL(k_ldl_update_rs2<T>, dim3(4 * (sa + sb)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, ua, sa, ub, sb, panel, c2a, c2b);
        BA_HIP_CHECK(hipEventRecord(w->ev_recv[slot], st));
        BA_HIP_CHECK(hipStreamWaitEvent(w->rest, w->ev_recv[slot], 0));
        const RunPanel<T> qa2 = panel_of(ka, VA, w->prow + la + 1 + nla), qb2 = panel_of(kb, VB, w->prow + lb + 1 + nlb);
        const int ntile = rest_a + rest_b;
        hipLaunchKernelGGL(k_ldl_update_part2<T>, dim3((ntile + 1) / 2 < rest_cus ? (ntile + 1) / 2 : rest_cus), dim3(512), part_lds_bytes<T>(), w->rest, w->S, w->col_off,
                           qa2, rest_a, qb2, rest_b, panel);
        BA_HIP_CHECK(hipEventRecord(w->ev_upd[slot], w->rest));
        pending[slot] = true;
      } else {
        BA_CHECK(join_rest(slot ^ 1));
        const RunPanel<T> ua = panel_of(ka, VA, w->prow + la + 1), ub = panel_of(kb, VB, w->prow + lb + 1);
        const int na = c2a * (c2a + 1) / 2, nb2 = c2b * (c2b + 1) / 2;
        if (na + nb2 <= update_rs_max())
          hipLaunchKernelGGL(k_ldl_update_rs2<T>, dim3(4 * (na + nb2)), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, ua, na, ub, nb2, panel, 0, 0);
        else
          hipLaunchKernelGGL(k_ldl_update2<T>, dim3(((na + 7) / 8) * 8 + ((nb2 + 7) / 8) * 8), dim3(256), gemm_priv_lds_bytes<T>(), st, w->S, w->col_off,
                             ua, na, ub, nb2, panel);
      }
    }
    BA_CHECK(join_rest(0));
    BA_CHECK(join_rest(1));
    BA_HIP_CHECK(hipGetLastError());
    for (int q = both; q < pat->a_clean; q++) order.push_back(q);  // the longer run's remainder, then everything else
    for (int q = pat->a_clean; q < pat->split; q++) order.push_back(q);
    for (int q = pat->split + both; q < npairs; q++) order.push_back(q);
  } else {
    for (int q = 0; q < npairs; q++) order.push_back(q);
  }
  for (size_t it = 0; it < order.size(); it++) {
    const int q = order[it], k = 2 * q;
    const int slot = (int)(it & 1);
    T *V0 = w->V + 2 * slot * panel, *V1 = V0 + panel;
    const int l0 = pat->prow_ptr[(size_t)q];
    const int *rows2 = w->prow + l0 + 1;
    BA_CHECK(join_rest(slot));  // (the rest of the pair two steps back read these panel buffers; joined long ago: see lead below)
    const int c2 = chain(q, st, V0, V1, !two && q == 0);  // (in one chain the first diagonal kernel clears the pivot flag)
    if (c2 == 0) continue;
    // the rows of U_q that are the next pair's own tile columns (k+2, k+3): at the head of the ascending list
    const bool next_adjacent = it + 1 < order.size() && order[it + 1] == q + 1;
    int nlead = 0;
    while (nlead < c2 && nlead < 2 && pat->prow[(size_t)(l0 + 1 + nlead)] < k + 4) nlead++;
    const int nrest = c2 - nlead;
    if (lookahead && next_adjacent && nrest * (nrest + 1) / 2 >= la_min) {
      // the lead first, alone on the chip (beside the rest it takes as long as the whole update: measured), then the rest
      // beside the next chain
      BA_CHECK(join_rest(slot ^ 1));  // rest(q-1) has updated the lead tiles too (and the next chain reads its columns)
      if (nlead > 0) {
        const int nblk = nlead == 1 ? c2 : 2 * c2 - 1;
        ProfScope ps(p, PC_LDL_UPDATE_RS, st);
        hipLaunchKernelGGL(k_ldl_update_rs<T>, dim3(4 * nblk), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, V0, V1, k, k + 2,
                           nblk, rows2, c2);
      }
      BA_HIP_CHECK(hipEventRecord(w->ev_recv[slot], st));  // both panels of pair q complete, its lead tiles updated
      BA_HIP_CHECK(hipStreamWaitEvent(w->rest, w->ev_recv[slot], 0));
      {
        const int nblk = nrest * (nrest + 1) / 2;
        hipLaunchKernelGGL(k_ldl_update_part<T>, dim3((nblk + 1) / 2 < rest_cus ? (nblk + 1) / 2 : rest_cus), dim3(512), part_lds_bytes<T>(), w->rest, w->S, w->col_off,
                           V0, V1, k, nblk, rows2 + nlead);
      }
      BA_HIP_CHECK(hipEventRecord(w->ev_upd[slot], w->rest));
      pending[slot] = true;
    } else {
      BA_CHECK(join_rest(slot ^ 1));
      launch_update(st, V0, V1, k, rows2, c2);
    }
  }
  BA_CHECK(join_rest(0));
  BA_CHECK(join_rest(1));
  BA_HIP_CHECK(hipGetLastError());
  if (zero_pivot) {
    int h = 0;
    BA_HIP_CHECK(hipMemcpyAsync(&h, w->flag, sizeof(int), hipMemcpyDeviceToHost, st));
    BA_HIP_CHECK(hipStreamSynchronize(st));
    *zero_pivot = h;
  }
  return BA_OK;
}

// pair update (panels k, k+1) of the owned tile columns own_cols[m0 .. m1)
template <typename T>
static int launch_pair_owned(ba_problem *p, DenseLDLT<T> *w, int k, const T *V0, const T *V1, hipStream_t st, int m0, int m1) {
  if (m0 >= m1) return BA_OK;
  const int64_t nblk64 = w->h_own_pref[(size_t)m1] - w->h_own_pref[(size_t)m0];
  if (nblk64 <= 0) return BA_OK;
  const int nblk = (int)nblk64;
  ProfScope ps(p, PC_LDL_UPDATE, st);
  const T *Lp0 = nullptr, *Lp1 = nullptr;
  if (w->own_only) {  // the panels' L tiles: the buffer that travels with the V buffer this pair uses
    Lp0 = w->Lb + (V0 - w->V);
    Lp1 = w->Lb + (V1 - w->V);
  }
  hipLaunchKernelGGL((k_ldl_update<T, 1, true>), dim3(((nblk + 7) / 8) * 8), dim3(256), gemm_priv_lds_bytes<T>(), st,
                     w->S, w->col_off, V0, V1, k, w->h_own_cols[(size_t)m0], (int)w->nt, nblk, (int *)nullptr, w->own_cols, w->own_pref, m0, m1,
                     1, (const int *)nullptr, Lp0, Lp1);
  return BA_OK;
}

// the tiles [from, to) of pair q's update list (block-sparse S on several ranks: the pattern's tiles in this rank's columns)
template <typename T>
static int launch_pair_list(ba_problem *p, DenseLDLT<T> *w, int q, const T *V0, const T *V1, hipStream_t st, int from, int to) {
  const int nblk = to - from;
  if (nblk <= 0) return BA_OK;
  ProfScope ps(p, PC_LDL_UPDATE, st);
  hipLaunchKernelGGL((k_ldl_update<T, 1>), dim3(((nblk + 7) / 8) * 8), dim3(256), gemm_priv_lds_bytes<T>(), st, w->S, w->col_off, V0, V1,
                     2 * q, 2 * q + 2, (int)w->nt, nblk, (int *)nullptr, (const int *)nullptr, (const int64_t *)nullptr, 0, 0, 1,
                     (const int *)nullptr, (const T *)(w->Lb + (V0 - w->V)), (const T *)(w->Lb + (V1 - w->V)),
                     (const int2 *)(w->upd_ij + w->h_upd_ptr[(size_t)q] + from));
  return BA_OK;
}

// the panel chain of pair (k, k+1) on its owner (dense_ldl_factor's in-order chain, no right-hand side)
template <typename T>
static int dist_chain(ba_problem *p, DenseLDLT<T> *w, int k, T *V0, T *V1, hipStream_t st) {
  if (w->sparse) {  // over the pattern's row lists, as dense_ldl_factor_sparse
    const TilePattern *pat = w->pat;
    const int q = k / 2, l0 = pat->prow_ptr[(size_t)q], c1 = pat->prow_ptr[(size_t)q + 1] - l0, c2 = c1 > 0 ? c1 - 1 : 0;
    launch_diag(p, w, k, st);
    if (c1 > 0) {
      {
        ProfScope ps(p, PC_LDL_TRSM, st);
        hipLaunchKernelGGL((k_ldl_trsm_rs<T, false>), dim3(4 * c1), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off,
                           w->Linv + (int64_t)k * NB * NB, w->D + (int64_t)k * NB, V0, k, (T *)nullptr, (T *)nullptr, w->prow + l0);
      }
      {
        ProfScope ps(p, PC_LDL_SYRK, st);
        hipLaunchKernelGGL(k_ldl_col_rs<T>, dim3(4 * c1), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off, V0, k, w->prow + l0);
      }
      launch_diag(p, w, k + 1, st);
      if (c2 > 0) {
        ProfScope ps(p, PC_LDL_TRSM, st);
        hipLaunchKernelGGL((k_ldl_trsm_rs<T, false>), dim3(4 * c2), dim3(256), RS_LDS_ELEMS * sizeof(T), st, w->S, w->col_off,
                           w->Linv + (int64_t)(k + 1) * NB * NB, w->D + (int64_t)(k + 1) * NB, V1, k + 1, (T *)nullptr, (T *)nullptr,
                           w->prow + l0 + 1);
      }
    }
    BA_HIP_CHECK(hipGetLastError());
    return BA_OK;
  }
  launch_diag(p, w, k, st);
  launch_trsm(p, w, k, V0, (T *)nullptr, st);
  if (k + 1 < (int)w->nt) {
    launch_col(p, w, k, V0, st);
    launch_diag(p, w, k + 1, st);
    launch_trsm(p, w, k + 1, V1, (T *)nullptr, st);
  }
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// pair (k, k+1) from its owner to everybody: V = L D of both panels, the two inverted diagonal tiles, the 256 pivots (one
// grouped broadcast); the receivers rebuild L = V D^-1 in their copy of S
template <typename T>
static int dist_transfer(ba_problem *p, DenseLDLT<T> *w, int k, T *V0, T *V1, int owner, hipStream_t st) {
  const int nt = (int)w->nt;
  const bool two = k + 1 < nt;
  if (w->sparse) {  // the pattern's tile rows only: one broadcast per run of consecutive rows (a band: one run per panel)
    const TilePattern *pat = w->pat;
    const int q = k / 2, l0 = pat->prow_ptr[(size_t)q], c1 = pat->prow_ptr[(size_t)q + 1] - l0, c2 = c1 > 0 ? c1 - 1 : 0;
    BA_CHECK(comm_group_begin(p));
    int rc = BA_OK;
    auto bcast_rows = [&](T *V, int first, int cnt) {
      for (int a = 0; a < cnt && rc == BA_OK;) {
        int b = a + 1;
        while (b < cnt && pat->prow[(size_t)(first + b)] == pat->prow[(size_t)(first + b - 1)] + 1) b++;
        rc = comm_bcast(p, V + (int64_t)pat->prow[(size_t)(first + a)] * NB * NB, (int64_t)(b - a) * NB * NB * sizeof(T), owner, st);
        a = b;
      }
    };
    bcast_rows(V0, l0, c1);
    if (two) bcast_rows(V1, l0 + 1, c2);
    if (rc == BA_OK) rc = comm_bcast(p, w->Linv + (int64_t)k * NB * NB, (int64_t)(two ? 2 : 1) * NB * NB * sizeof(T), owner, st);
    if (rc == BA_OK) rc = comm_bcast(p, w->D + (int64_t)k * NB, (int64_t)(two ? 2 : 1) * NB * sizeof(T), owner, st);
    BA_CHECK(comm_group_end(p));
    BA_CHECK(rc);
    T *L0 = w->Lb + (V0 - w->V), *L1 = w->Lb + (V1 - w->V);
    if (c1 > 0) hipLaunchKernelGGL(k_ldl_scale_panel<T>, dim3(c1), dim3(256), 0, st, L0, V0, w->D + (int64_t)k * NB, 0, w->prow + l0);
    if (two && c2 > 0)
      hipLaunchKernelGGL(k_ldl_scale_panel<T>, dim3(c2), dim3(256), 0, st, L1, V1, w->D + (int64_t)(k + 1) * NB, 0, w->prow + l0 + 1);
    BA_HIP_CHECK(hipGetLastError());
    return BA_OK;
  }
  const int rows0 = nt - k - 1, rows1 = nt - k - 2;  // tile rows below the diagonal tile of column k / k + 1
  BA_CHECK(comm_group_begin(p));
  int rc = BA_OK;
  if (rows0 > 0) rc = comm_bcast(p, V0 + (int64_t)(k + 1) * NB * NB, (int64_t)rows0 * NB * NB * sizeof(T), owner, st);
  if (rc == BA_OK && two && rows1 > 0)
    rc = comm_bcast(p, V1 + (int64_t)(k + 2) * NB * NB, (int64_t)rows1 * NB * NB * sizeof(T), owner, st);
  if (rc == BA_OK) rc = comm_bcast(p, w->Linv + (int64_t)k * NB * NB, (int64_t)(two ? 2 : 1) * NB * NB * sizeof(T), owner, st);
  if (rc == BA_OK) rc = comm_bcast(p, w->D + (int64_t)k * NB, (int64_t)(two ? 2 : 1) * NB * sizeof(T), owner, st);
  BA_CHECK(comm_group_end(p));
  BA_CHECK(rc);
  if (w->own_only) {
    // every rank (the owner too: one source for the update's operand) rebuilds L = V D^-1 of both panels in the panel
    // buffer beside V; nothing of another rank's columns enters S
    T *L0 = w->Lb + (V0 - w->V), *L1 = w->Lb + (V1 - w->V);
    if (rows0 > 0) hipLaunchKernelGGL(k_ldl_scale_panel<T>, dim3(rows0), dim3(256), 0, st, L0, V0, w->D + (int64_t)k * NB, k + 1);
    if (two && rows1 > 0)
      hipLaunchKernelGGL(k_ldl_scale_panel<T>, dim3(rows1), dim3(256), 0, st, L1, V1, w->D + (int64_t)(k + 1) * NB, k + 2);
    BA_HIP_CHECK(hipGetLastError());
  } else if (owner != w->rank) {
    if (rows0 > 0)
      hipLaunchKernelGGL(k_ldl_scale_panel<T>, dim3(rows0), dim3(256), 0, st, w->S + (w->hco()[k] - k) * NB * NB, V0,
                         w->D + (int64_t)k * NB, k + 1);
    if (two && rows1 > 0)
      hipLaunchKernelGGL(k_ldl_scale_panel<T>, dim3(rows1), dim3(256), 0, st, w->S + (w->hco()[k + 1] - (k + 1)) * NB * NB, V1,
                         w->D + (int64_t)(k + 1) * NB, k + 2);
    BA_HIP_CHECK(hipGetLastError());
  }
  return BA_OK;
}

// per-rank ownership: the forward substitution of the right-hand side rides along with the panels as they arrive -- every
// rank holds the pair's L tiles (panel buffer) exactly then, and never again.  Replicated: every rank does the same
// arithmetic on its copy of b.
template <typename T>
static int dist_forward_pair(DenseLDLT<T> *w, int k, T *V0, T *V1, T *d_b, hipStream_t st) {
  if (!d_b) return BA_OK;
  const int nt = (int)w->nt;
  T *y = w->D + (int64_t)nt * NB;
  if (w->sparse) {
    const TilePattern *pat = w->pat;
    const int q = k / 2, l0 = pat->prow_ptr[(size_t)q], c1 = pat->prow_ptr[(size_t)q + 1] - l0, c2 = c1 > 0 ? c1 - 1 : 0;
    hipLaunchKernelGGL(k_fwd_step<T>, dim3(1 + c1), dim3(256), 0, st, w->S, w->col_off, w->Linv, d_b, y, k, (const int *)(w->prow + l0),
                       (const T *)(w->Lb + (V0 - w->V)));
    if (k + 1 < nt)
      hipLaunchKernelGGL(k_fwd_step<T>, dim3(1 + c2), dim3(256), 0, st, w->S, w->col_off, w->Linv, d_b, y, k + 1,
                         (const int *)(w->prow + l0 + 1), (const T *)(w->Lb + (V1 - w->V)));
    BA_HIP_CHECK(hipGetLastError());
    return BA_OK;
  }
  hipLaunchKernelGGL(k_fwd_step<T>, dim3(nt - k), dim3(256), 0, st, w->S, w->col_off, w->Linv, d_b, y, k, (const int *)nullptr,
                     (const T *)(w->Lb + (V0 - w->V)));
  if (k + 1 < nt)
    hipLaunchKernelGGL(k_fwd_step<T>, dim3(nt - k - 1), dim3(256), 0, st, w->S, w->col_off, w->Linv, d_b, y, k + 1, (const int *)nullptr,
                       (const T *)(w->Lb + (V1 - w->V)));
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// Distributed right-looking factorisation over the ranks of p->comm: tile column pair q = (2q, 2q+1) belongs to rank
// q % world.  Per pair: the owner runs the panel chain of dense_ldl_factor on its (fully updated) columns, broadcasts
// V = L D of both panels, the two inverted diagonal tiles and the 256 pivots; the other ranks rebuild L = V D^-1 in their
// copy of S; every rank then applies the pair update (K = 256) to the tile columns it owns.  On exit every rank holds the
// whole factor, so the triangular solves are replicated and bit-identical everywhere.  Traffic per rank: the panels,
// n^2/2 elements in total (Venice, n = 16 002: 1.0 GB; Final-13682, n = 123 138: 60 GB Float64 / 30 GB Float32) against
// n^3/(3 world) flops.  The hoisted-diagonal schedule is a single-GPU refinement and is not used here.
//
// Look-ahead of one pair (default; BA_DIST_LOOKAHEAD=0 gives the strictly alternating chain / broadcast / update): the
// owner of pair q+1 applies pair q's update to ITS two leading tile columns first, runs the chain of pair q+1 at once and
// hands the panels to the transfer stream, then finishes its share of update q; everybody else receives pair q+1 on the
// transfer stream while update q runs.  Chain and broadcast of pair q+1 thus hide behind update q wherever that update is
// the longer of the two (Final-13682 on 8 ranks: 2.4 ms of update per pair and rank against ~0.7 ms of chain and ~2 ms of
// broadcast).  The arithmetic -- which tile receives which products in which order -- is that of the alternating
// schedule: the results are bit-identical (tests/test_distributed.py).  Buffers: pair q's panels live in Vs[q & 1];
// transfer q+1 may overwrite Vs[(q+1) & 1] only when update q-1 is through with it (ev_upd), update q+1 starts when
// transfer q+1 has landed (ev_recv).
template <typename T>
int dense_ldl_factor_dist(ba_problem *p, DenseLDLT<T> *w, hipStream_t st, T *d_b) {
  if (d_b && !w->own_only) {
    ba_set_error("dense_ldl_factor_dist: the fused forward substitution belongs to the per-rank-ownership layout");
    return BA_ERR_ARG;
  }
  const int nt = (int)w->nt, P = w->world, me = w->rank;
  const int64_t panel = (int64_t)nt * NB * NB;
  T *Vs[2][2] = {{w->V, w->V + panel}, {w->V + 2 * panel, w->V + 3 * panel}};
  BA_HIP_CHECK(hipMemsetAsync(w->flag, 0, sizeof(int), st));
  w->hoisting = false;
  const char *la_env = getenv("BA_DIST_LOOKAHEAD");  // read per call: a test flips it between two factorisations of one process
  const bool la_off = la_env && la_env[0] == '0';
  const int M = (int)w->h_own_cols.size();
  auto own_from = [&](int base) {  // index of the first owned tile column >= base
    return (int)(std::lower_bound(w->h_own_cols.begin(), w->h_own_cols.end(), base) - w->h_own_cols.begin());
  };
  if (w->sparse && !w->own_only) {
    ba_set_error("dense_ldl_factor_dist: a tile pattern needs per-rank ownership of S");
    return BA_ERR_ARG;
  }
  // pair q's update of this rank's tile columns: `part` 0 = all of it, 1 = the next pair's own two columns only (when they
  // are this rank's), 2 = what remains after part 1
  auto update_mine = [&](int q, const T *V0, const T *V1, int part, bool next_mine) -> int {
    const int k = 2 * q;
    if (w->sparse) {
      const int n = w->h_upd_ptr[(size_t)q + 1] - w->h_upd_ptr[(size_t)q], lead = next_mine ? w->h_upd_lead[(size_t)q] : 0;
      if (part == 1) return launch_pair_list(p, w, q, V0, V1, st, 0, lead);
      return launch_pair_list(p, w, q, V0, V1, st, part == 2 ? lead : 0, n);
    }
    int m0 = own_from(k + 2);
    const int lead = next_mine ? ((k + 3 < nt) ? 2 : 1) : 0;
    if (part == 1) return launch_pair_owned(p, w, k, V0, V1, st, m0, m0 + lead);
    return launch_pair_owned(p, w, k, V0, V1, st, part == 2 ? m0 + lead : m0, M);
  };
  if (la_off || p->prof_on) {  // (per-kernel profiling times one launch at a time: nothing could overlap)
    for (int k = 0, q = 0; k < nt; k += 2, q++) {
      T *V0 = Vs[q & 1][0], *V1 = Vs[q & 1][1];
      const int owner = q % P;
      if (owner == me) BA_CHECK(dist_chain(p, w, k, V0, V1, st));
      BA_CHECK(dist_transfer(p, w, k, V0, V1, owner, st));
      BA_CHECK(dist_forward_pair(w, k, V0, V1, d_b, st));
      if (k + 2 >= nt) break;
      BA_CHECK(update_mine(q, V0, V1, 0, false));
    }
  } else {
    hipStream_t cs = w->hoist;  // transfer stream
    BA_HIP_CHECK(hipEventRecord(w->ev_dtop, st));  // fork: behind the reduce of S
    BA_HIP_CHECK(hipStreamWaitEvent(cs, w->ev_dtop, 0));
    if (0 % P == me) {
      BA_CHECK(dist_chain(p, w, 0, Vs[0][0], Vs[0][1], st));
      BA_HIP_CHECK(hipEventRecord(w->ev_dchain, st));
      BA_HIP_CHECK(hipStreamWaitEvent(cs, w->ev_dchain, 0));
    }
    BA_CHECK(dist_transfer(p, w, 0, Vs[0][0], Vs[0][1], 0, cs));
    BA_HIP_CHECK(hipEventRecord(w->ev_recv[0], cs));
    for (int k = 0, q = 0; k < nt; k += 2, q++) {
      T *V0 = Vs[q & 1][0], *V1 = Vs[q & 1][1];
      T *N0 = Vs[(q + 1) & 1][0], *N1 = Vs[(q + 1) & 1][1];
      BA_HIP_CHECK(hipStreamWaitEvent(st, w->ev_recv[q & 1], 0));  // pair q's panels are here (join of the transfer stream)
      BA_CHECK(dist_forward_pair(w, k, V0, V1, d_b, st));  // (ahead of update q, whose event frees this buffer for transfer q+2)
      if (k + 2 >= nt) break;
      const bool next_mine = (q + 1) % P == me;
      if (next_mine) {  // tile columns k+2 [, k+3] are mine and come first in the owned list
        BA_CHECK(update_mine(q, V0, V1, 1, true));
        BA_CHECK(dist_chain(p, w, k + 2, N0, N1, st));
        BA_HIP_CHECK(hipEventRecord(w->ev_dchain, st));
      }
      BA_CHECK(update_mine(q, V0, V1, 2, next_mine));
      BA_HIP_CHECK(hipEventRecord(w->ev_upd[q & 1], st));
      if (q >= 1) BA_HIP_CHECK(hipStreamWaitEvent(cs, w->ev_upd[(q + 1) & 1], 0));  // update q-1 has read Vs[(q+1)&1]
      if (next_mine) BA_HIP_CHECK(hipStreamWaitEvent(cs, w->ev_dchain, 0));
      BA_CHECK(dist_transfer(p, w, k + 2, N0, N1, (q + 1) % P, cs));
      BA_HIP_CHECK(hipEventRecord(w->ev_recv[(q + 1) & 1], cs));
    }
  }
  // an exactly zero pivot is seen by the owner of that tile only: make the flag collective
  hipLaunchKernelGGL(k_flag_to_double, dim3(1), dim3(1), 0, st, w->flag, w->flag_sum);
  BA_CHECK(comm_allreduce(p, w->flag_sum, 1, st));
  hipLaunchKernelGGL(k_double_to_flag, dim3(1), dim3(1), 0, st, w->flag_sum, w->flag);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// ---- backward sweep with per-rank ownership of S ---------------------------------------------------------------------------
// A rank holds whole tile COLUMNS of L, so the sweep runs column-wise from the last pair to the first: the owner of pair
// (k, k+1) forms s_c = sum_{i > c} L_ic' x_i for its two columns from x of the later rows (replicated: every rank has
// received them), x_{k+1} = Linv_{k+1}' (D^-1 y_{k+1} - s_{k+1}), x_k likewise with the coupling tile L_{k+1,k}, and
// broadcasts the 256 values.  Per pair: one row-parallel kernel (a block per tile below the pair, partial products in a
// fixed order), one finishing workgroup, one 2 KB broadcast; every rank ends with the same x.  (The forward substitution
// rode along with the panels, dist_forward_pair.)
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_col_part(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ x,
                                                       T *__restrict__ part, int k, int nt, const int *__restrict__ rows = nullptr) {
  // block b: tile row i = k + 2 + b / 2 (rows: the b / 2-th tile row of the pair's pattern) of column c = k + (b & 1):
  // part[(c - k) nt + i][:] = L_ic' x_i
  __shared__ T xi[NB], red[2][NB];
  const int c = k + (blockIdx.x & 1), i = rows ? rows[blockIdx.x >> 1] : k + 2 + (blockIdx.x >> 1);
  const int tid = threadIdx.x, col = tid & (NB - 1), half = tid >> 7;
  if (tid < NB) xi[tid] = x[(int64_t)i * NB + tid];
  __syncthreads();
  const T *L = S + tix(co, i, c) * NB * NB;
  T s = 0;
  for (int r = half * 64; r < half * 64 + 64; r++) s += L[r * NB + col] * xi[r];
  red[half][col] = s;
  __syncthreads();
  if (tid < NB) part[((int64_t)(c - k) * nt + i) * NB + tid] = red[0][tid] + red[1][tid];
}

template <typename T>
__global__ __launch_bounds__(256) void k_bwd_col_final(const T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ Linv,
                                                        const T *__restrict__ D, const T *__restrict__ y, const T *__restrict__ part,
                                                        T *__restrict__ x, int k, int nt, const int *__restrict__ rows = nullptr,
                                                        int nrows = 0) {
  // rows / nrows: the tile rows below the pair that its pattern holds (block-sparse S); null: k + 2 .. nt - 1
  __shared__ T z[NB], xk1[NB], red[2][NB];
  const int tid = threadIdx.x, col = tid & (NB - 1), half = tid >> 7;
  const bool two = k + 1 < nt;
  auto matvec_t = [&](const T *M, const T *v) {  // red[half][col] = sum over this half's 64 rows of M[r][col] v[r]
    T s = 0;
    for (int r = half * 64; r < half * 64 + 64; r++) s += M[r * NB + col] * v[r];
    red[half][col] = s;
  };
  if (two) {  // x_{k+1} = Linv_{k+1}' (y_{k+1} / D_{k+1} - sum_{i > k+1} L_{i,k+1}' x_i)
    if (tid < NB) {
      T s = 0;
      if (rows)
        for (int a = 0; a < nrows; a++) s += part[((int64_t)nt + rows[a]) * NB + tid];
      else
        for (int i = k + 2; i < nt; i++) s += part[((int64_t)nt + i) * NB + tid];
      z[tid] = y[(int64_t)(k + 1) * NB + tid] / D[(int64_t)(k + 1) * NB + tid] - s;
    }
    __syncthreads();
    matvec_t(Linv + (int64_t)(k + 1) * NB * NB, z);
    __syncthreads();
    if (tid < NB) {
      xk1[tid] = red[0][tid] + red[1][tid];
      x[(int64_t)(k + 1) * NB + tid] = xk1[tid];
    }
    __syncthreads();
    matvec_t(S + tix(co, k + 1, k) * NB * NB, xk1);  // the coupling tile L_{k+1,k}' x_{k+1}
    __syncthreads();
  }
  if (tid < NB) {
    T s = two ? red[0][tid] + red[1][tid] : (T)0;
    if (rows)
      for (int a = 0; a < nrows; a++) s += part[(int64_t)rows[a] * NB + tid];
    else
      for (int i = k + 2; i < nt; i++) s += part[(int64_t)i * NB + tid];
    z[tid] = y[(int64_t)k * NB + tid] / D[(int64_t)k * NB + tid] - s;
  }
  __syncthreads();
  matvec_t(Linv + (int64_t)k * NB * NB, z);
  __syncthreads();
  if (tid < NB) x[(int64_t)k * NB + tid] = red[0][tid] + red[1][tid];
}

template <typename T>
static int dense_ldl_bwd_dist(ba_problem *p, DenseLDLT<T> *w, T *d_b, hipStream_t st) {
  const int nt = (int)w->nt, P = w->world, me = w->rank;
  const T *y = w->D + (int64_t)nt * NB;
  const int last = ((nt - 1) / 2) * 2;
  for (int k = last; k >= 0; k -= 2) {
    const int owner = (k / 2) % P;
    const bool two = k + 1 < nt;
    if (owner == me && w->sparse) {  // the pair's pattern rows only
      const TilePattern *pat = w->pat;
      const int q = k / 2, l0 = pat->prow_ptr[(size_t)q], c1 = pat->prow_ptr[(size_t)q + 1] - l0, c2 = c1 > 0 ? c1 - 1 : 0;
      if (c2 > 0)
        hipLaunchKernelGGL(k_bwd_col_part<T>, dim3(2 * c2), dim3(256), 0, st, w->S, w->col_off, d_b, w->bpart, k, nt, (const int *)(w->prow + l0 + 1));
      hipLaunchKernelGGL(k_bwd_col_final<T>, dim3(1), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, w->bpart, d_b, k, nt,
                         (const int *)(w->prow + l0 + 1), c2);
      BA_HIP_CHECK(hipGetLastError());
    } else if (owner == me) {
      const int below = nt - k - 2;  // tile rows below the pair
      if (below > 0)
        hipLaunchKernelGGL(k_bwd_col_part<T>, dim3(2 * below), dim3(256), 0, st, w->S, w->col_off, d_b, w->bpart, k, nt);
      hipLaunchKernelGGL(k_bwd_col_final<T>, dim3(1), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, w->bpart, d_b, k, nt);
      BA_HIP_CHECK(hipGetLastError());
    }
    BA_CHECK(comm_bcast(p, d_b + (int64_t)k * NB, (int64_t)(two ? 2 : 1) * NB * sizeof(T), owner, st));
  }
  return BA_OK;
}

template <typename T>
int dense_ldl_solve(ba_problem *p, DenseLDLT<T> *w, T *d_b, hipStream_t st, bool forward_done) {
  const int nt = (int)w->nt;
  T *y = w->D + (int64_t)nt * NB;
  ProfScope ps(p, PC_SOLVE, st);
  if (w->own_only && p->comm.active()) {
    if (!forward_done) {
      ba_set_error("per-rank ownership of S: the forward substitution rides along with the distributed factorisation");
      return BA_ERR_ARG;
    }
    return dense_ldl_bwd_dist(p, w, d_b, st);
  }
  if (!forward_done)
    for (int k = 0; k < nt; k++)
      hipLaunchKernelGGL(k_fwd_step<T>, dim3(nt - k), dim3(256), 0, st, w->S, w->col_off, w->Linv, d_b, y, k);
  if (w->sparse && !p->comm.active()) {  // block-sparse S: one tile row per launch over the tile columns of its pattern
    const TilePattern *pat = w->pat;
    if (!forward_done) {
      ba_set_error("block-sparse reduced camera system: the forward substitution rides along with the factorisation");
      return BA_ERR_ARG;
    }
    // The tile rows of one tile column pair (2q + 1, 2q) per launch over the union of their pattern columns (a last single row
    // when nt is odd), last pair first.  Two groups of columns (TilePattern::split / b_clean: the ordering eliminated a
    // profile from both ends): the rows [2 split, 2 (split + b_clean)) have no pattern column before the split -- no pair of the
    // first group touches them -- and the rows before the split none behind it, so once the rows behind both groups are done
    // the two groups' sweeps are independent: two pairs per launch (k_bwd_pair2).  Every y_j receives its updates in the same
    // sequence as in the one-pair-per-launch sweep: same bits.
    const int npairs = (nt + 1) / 2;
    auto single = [&](int q) {
      const int k = 2 * q + 1;
      if (k >= nt) {
        const int c0 = pat->lcol_ptr[(size_t)(nt - 1)], c1 = pat->lcol_ptr[(size_t)nt];
        hipLaunchKernelGGL(k_bwd_step<T>, dim3(1 + (c1 - c0)), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, d_b, nt - 1, w->lcol + c0);
      } else {
        const int c0 = pat->lpair_ptr[(size_t)q], c1 = pat->lpair_ptr[(size_t)q + 1];
        hipLaunchKernelGGL(k_bwd_pair<T>, dim3(1 + (c1 - c0)), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, d_b, k, w->lpair + c0);
      }
    };
    const char *tc_env = getenv("BA_SPARSE_TWO_RUNS");  // (read per call: a test compares the sweeps in one process)
    int c_hi = pat->split + pat->b_clean;
    if (c_hi > 0 && 2 * c_hi - 1 >= nt) c_hi--;  // (a last single row stays with the rows behind the groups)
    const char *b2_env = getenv("BA_SPARSE_BWD2");
    if (pat->split > 0 && c_hi > pat->split && !(tc_env && tc_env[0] == '0') && !(b2_env && b2_env[0] == '0')) {
      for (int q = npairs - 1; q >= c_hi; q--) single(q);
      const int both = std::min(pat->split, c_hi - pat->split);
      for (int i = 0; i < both; i++) {
        const int qa = pat->split - 1 - i, qc = c_hi - 1 - i;
        const int a0 = pat->lpair_ptr[(size_t)qa], a1 = pat->lpair_ptr[(size_t)qa + 1];
        const int c0 = pat->lpair_ptr[(size_t)qc], c1 = pat->lpair_ptr[(size_t)qc + 1];
        hipLaunchKernelGGL(k_bwd_pair2<T>, dim3(2 + (a1 - a0) + (c1 - c0)), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, d_b,
                           2 * qa + 1, w->lpair + a0, 1 + (a1 - a0), 2 * qc + 1, w->lpair + c0);
      }
      for (int q = pat->split - 1 - both; q >= 0; q--) single(q);
      for (int q = c_hi - 1 - both; q >= pat->split; q--) single(q);
    } else {
      for (int q = npairs - 1; q >= 0; q--) single(q);
    }
    BA_HIP_CHECK(hipGetLastError());
    return BA_OK;
  }
  static const bool pair_off = [] { const char *e = getenv("BA_BWD_PAIR"); return e && e[0] == '0'; }();
  int k = nt - 1;
  if (!pair_off)
    for (; k >= 1; k -= 2)  // panels k, k-1 per launch; blocks: 1 + (k-1) tile rows below the pair
      hipLaunchKernelGGL(k_bwd_pair<T>, dim3(k), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, d_b, k);
  for (; k >= 0; k--)
    hipLaunchKernelGGL(k_bwd_step<T>, dim3(k + 1), dim3(256), 0, st, w->S, w->col_off, w->Linv, w->D, y, d_b, k);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// ---- C ABI: standalone dense solve (tests, roofline measurement) --------------------------------------------------------
template <typename T>
static int dense_solve_host(int device, int64_t n, const double *a_lower_rowmajor, const double *b, double *x,
                            double *factor_ms) {
  if (n <= 0 || !a_lower_rowmajor || !b || !x) {
    ba_set_error("ba_dense_ldl_solve: bad argument");
    return BA_ERR_ARG;
  }
  BA_HIP_CHECK(hipSetDevice(device));
  ba_problem tmp;  // only used for its (disabled) profiling slots
  DenseLDLT<T> w;
  int rc = dense_ldl_alloc<T>(&w, n, nullptr);
  if (rc != BA_OK) return rc;
  const int64_t npad = w.n;
  std::vector<T> tiles((size_t)dense_ldl_tiles_doubles(n), (T)0);
  for (int64_t i = 0; i < npad; i++) {
    int64_t ti = i / NB;
    for (int64_t j = 0; j <= i; j++) {
      int64_t tj = j / NB;
      double v = (i < n) ? a_lower_rowmajor[i * n + j] : (i == j ? 1.0 : 0.0);
      tiles[(size_t)((tix(w.hco(), ti, tj) * NB + (i - ti * NB)) * NB + (j - tj * NB))] = (T)v;
    }
  }
  std::vector<T> bb((size_t)npad, (T)0);
  for (int64_t i = 0; i < n; i++) bb[(size_t)i] = (T)b[i];
  T *d_b = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipMalloc((void **)&d_b, (size_t)npad * sizeof(T)));
  BA_HIP_CHECK(hipMemcpy(w.S, tiles.data(), tiles.size() * sizeof(T), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipMemcpy(d_b, bb.data(), (size_t)npad * sizeof(T), hipMemcpyHostToDevice));
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  int zp = 0;
  const bool fused = getenv("BA_LDL_SEPARATE_FORWARD") == nullptr;
  for (int attempt = 0; attempt < 2; attempt++) {
    BA_HIP_CHECK(hipEventRecord(e0, st));
    rc = dense_ldl_factor<T>(&tmp, &w, st, nullptr, fused ? d_b : nullptr);
    BA_HIP_CHECK(hipEventRecord(e1, st));
    if (rc == BA_OK) rc = dense_ldl_solve<T>(&tmp, &w, d_b, st, fused);
    BA_HIP_CHECK(hipMemcpy(&zp, w.flag, sizeof(int), hipMemcpyDeviceToHost));
    BA_HIP_CHECK(hipDeviceSynchronize());
    if (zp != 2 || w.hoist_disabled) break;
    // a hoisted diagonal kernel gave up waiting (kernels serialised, e.g. under a counter-collecting profiler): redo in order
    w.hoist_disabled = true;
    BA_HIP_CHECK(hipMemcpy(w.S, tiles.data(), tiles.size() * sizeof(T), hipMemcpyHostToDevice));
    BA_HIP_CHECK(hipMemcpy(d_b, bb.data(), (size_t)npad * sizeof(T), hipMemcpyHostToDevice));
  }
  float ms = 0;
  BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (factor_ms) *factor_ms = ms;
  BA_HIP_CHECK(hipMemcpy(bb.data(), d_b, (size_t)npad * sizeof(T), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; i++) x[i] = (double)bb[(size_t)i];
  (void)hipFree(d_b);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dense_ldl_free<T>(&w);
  if (rc == BA_OK && zp) {
    ba_set_error("dense LDL': exactly zero pivot");
    return BA_ERR_ZERO_PIVOT;
  }
  return rc;
}

extern "C" int ba_dense_ldl_solve(int device, int64_t n, const double *a_lower_rowmajor, const double *b, double *x,
                                  double *factor_ms) {
  return dense_solve_host<double>(device, n, a_lower_rowmajor, b, x, factor_ms);
}

// same with the matrix and right-hand side rounded to Float32 and factored / solved in Float32 (facto_type = Float32)
extern "C" int ba_dense_ldl_solve_f32(int device, int64_t n, const double *a_lower_rowmajor, const double *b, double *x,
                                      double *factor_ms) {
  return dense_solve_host<float>(device, n, a_lower_rowmajor, b, x, factor_ms);
}

template int dense_ldl_alloc<double>(DenseLDLT<double> *, int64_t, double *, int, int, bool, bool);
template int dense_ldl_alloc<float>(DenseLDLT<float> *, int64_t, float *, int, int, bool, bool);
template int dense_ldl_alloc_S<double>(DenseLDLT<double> *);
template int dense_ldl_alloc_S<float>(DenseLDLT<float> *);
template void dense_ldl_free<double>(DenseLDLT<double> *);
template void dense_ldl_free<float>(DenseLDLT<float> *);
template int dense_ldl_factor<double>(ba_problem *, DenseLDLT<double> *, hipStream_t, int *, double *);
template int dense_ldl_factor<float>(ba_problem *, DenseLDLT<float> *, hipStream_t, int *, float *);
template int dense_ldl_use_pattern<double>(DenseLDLT<double> *, const TilePattern *);
template int dense_ldl_use_pattern<float>(DenseLDLT<float> *, const TilePattern *);
template int dense_ldl_factor_dist<double>(ba_problem *, DenseLDLT<double> *, hipStream_t, double *);
template int dense_ldl_factor_dist<float>(ba_problem *, DenseLDLT<float> *, hipStream_t, float *);
template int dense_ldl_solve<double>(ba_problem *, DenseLDLT<double> *, double *, hipStream_t, bool);
template int dense_ldl_solve<float>(ba_problem *, DenseLDLT<float> *, float *, hipStream_t, bool);
