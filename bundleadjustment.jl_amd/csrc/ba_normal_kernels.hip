// Normal-equation kernels of the Levenberg-Marquardt step (gfx950).
//
// The reference assembles K = [[I J];[J' -lambda I]] with sparse() every iteration and factors it with a
// scalar sparse LDL' (src/lm.jl:68-100,154-238; src/ldl_aux.jl:122-201).  Eliminating the residual rows and
// then the point columns of K in closed form gives the reduced camera system
//     S dc = rhs,  S = Hcc + lambda I - sum_p W_p' U_p^-1 W_p,  U_p = Hpp[p] + lambda I,
//     rhs = -(gc - sum_p W_p' U_p^-1 gp[p]),  dp = -U_p^-1 (gp + W_p dc)
// which is what these kernels build.  With the 2x12 block of observation a written [A_a | B_a]
// (A_a 2x3 point part, B_a 2x9 camera part):  W_a = A_a' B_a  and
//     W_a' U^-1 W_b = B_a' (A_a U^-1 A_b') B_b = B_a' Q_ab B_b     with Q_ab only 2x2,
// so nothing but J itself (24 doubles / observation, the jac_coord! layout) is ever stored.
//
// All sums are deterministic: point-side sums run over the point-sorted observation list (one lane per
// point), camera-side sums over the camera-sorted list (one workgroup per camera, fixed tree), the Schur
// blocks over a (camera_a, camera_b)-sorted task list (one wave per 9x9 block).  No float atomics.
#include <hip/hip_fp16.h>

#include "ba_internal.h"
#include "ba_lm_internal.h"

namespace {

constexpr int BLK = 256;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- point side: Hpp (xx,xy,xz,yy,yz,zz) and gp = A' r ------------------------------------------------
__global__ __launch_bounds__(BLK) void k_point_blocks(int64_t npnts, const int *__restrict__ pt_ptr,
                                                       const int *__restrict__ pt_obs, const double *__restrict__ J,
                                                       const double *__restrict__ r, double *__restrict__ Hpp,
                                                       double *__restrict__ gp) {
  int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= npnts) return;
  double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int q = pt_ptr[p]; q < pt_ptr[p + 1]; q++) {
    int64_t o = pt_obs[q];
    const double *Jo = J + 24 * o;
    double a0[3] = {Jo[0], Jo[1], Jo[2]}, a1[3] = {Jo[12], Jo[13], Jo[14]};
    h[0] += a0[0] * a0[0] + a1[0] * a1[0];
    h[1] += a0[0] * a0[1] + a1[0] * a1[1];
    h[2] += a0[0] * a0[2] + a1[0] * a1[2];
    h[3] += a0[1] * a0[1] + a1[1] * a1[1];
    h[4] += a0[1] * a0[2] + a1[1] * a1[2];
    h[5] += a0[2] * a0[2] + a1[2] * a1[2];
    if (gp) {
      double r0 = r[2 * o], r1 = r[2 * o + 1];
      g[0] += a0[0] * r0 + a1[0] * r1;
      g[1] += a0[1] * r0 + a1[1] * r1;
      g[2] += a0[2] * r0 + a1[2] * r1;
    }
  }
  if (Hpp)
    for (int i = 0; i < 6; i++) Hpp[6 * p + i] = h[i];
  if (gp)
    for (int i = 0; i < 3; i++) gp[3 * p + i] = g[i];
}

// ---- camera side: one workgroup per camera, fixed-order tree ----------------------------------------------
// MODE 0: Hcc (45, packed lower row-major) and gc = B' r (9).   MODE 1: gc only.
// MODE 2: rhs = sum_a B_a' (A_a u_p(a) - r_a)   (u = U^-1 gp per point).   MODE 3 (PCG product): Hcc_c x_c + lam x_c + sum_a B_a' A_a u_p(a), x in `r`.
template <int MODE>
__global__ __launch_bounds__(BLK) void k_cam_blocks(const int *__restrict__ cam_ptr, const int *__restrict__ cam_obs,
                                                     const int *__restrict__ pnt0, const double *__restrict__ J,
                                                     const double *__restrict__ r, const double *__restrict__ u,
                                                     double *__restrict__ Hcc, double *__restrict__ out9, double lam = 0.0,
                                                     const int *__restrict__ opos = nullptr) {
  // opos (MODE 2): the right-hand side of the camera system is written at the camera's block row of S (camera ordering)
  constexpr int NACC = (MODE == 0) ? 54 : 9;
  __shared__ double red[BLK / 64][NACC];
  const int c = blockIdx.x;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  for (int q = cam_ptr[c] + threadIdx.x; q < cam_ptr[c + 1]; q += BLK) {
    int64_t o = cam_obs[q];
    const double *Jo = J + 24 * o;
    double b0[9], b1[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      b0[i] = Jo[3 + i];
      b1[i] = Jo[15 + i];
    }
    double w0, w1;
    if (MODE == 2) {
      const double *up = u + 3 * (int64_t)pnt0[o];
      w0 = (Jo[0] * up[0] + Jo[1] * up[1] + Jo[2] * up[2]) - r[2 * o];
      w1 = (Jo[12] * up[0] + Jo[13] * up[1] + Jo[14] * up[2]) - r[2 * o + 1];
    } else if (MODE == 3) {
      const double *up = u + 3 * (int64_t)pnt0[o];
      w0 = Jo[0] * up[0] + Jo[1] * up[1] + Jo[2] * up[2];
      w1 = Jo[12] * up[0] + Jo[13] * up[1] + Jo[14] * up[2];
    } else {
      w0 = r[2 * o];
      w1 = r[2 * o + 1];
    }
    if (MODE == 0) {
      int idx = 0;
#pragma unroll
      for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) acc[idx++] += b0[i] * b0[j] + b1[i] * b1[j];
#pragma unroll
      for (int i = 0; i < 9; i++) acc[45 + i] += b0[i] * w0 + b1[i] * w1;
    } else {
#pragma unroll
      for (int i = 0; i < 9; i++) acc[i] += b0[i] * w0 + b1[i] * w1;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; i++) {
    double v = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    if (MODE == 0) {
      if (threadIdx.x < 45) Hcc[45 * (int64_t)c + threadIdx.x] = v;
      else out9[9 * (int64_t)c + threadIdx.x - 45] = v;
    } else if (MODE == 3) {  // PCG product: + Hcc_c x_c + lam x_c, x handed over in `r` (Hcc packed lower, read only)
      const int i = threadIdx.x;
      const double *x = r + 9 * (int64_t)c;
      const double *H = Hcc + 45 * (int64_t)c;
      double s = 0;
#pragma unroll
      for (int j = 0; j < 9; j++) s += H[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * x[j];
      out9[9 * (int64_t)c + i] = (s + v) + lam * x[i];
    } else {
      out9[9 * (int64_t)((MODE == 2 && opos) ? opos[c] : c) + threadIdx.x] = v;
    }
  }
}

// ---- coalesced row staging ------------------------------------------------------------------------------------------
// A lane that reads "its" observation's 24 doubles with scalar loads touches 64 different 192-byte rows per instruction.
// Here a wave stages the rows of 64 observations into its LDS slot with 16-byte loads in which 12 consecutive lanes read
// one whole row (12 instructions per 64 rows), and each lane then works on its row from LDS (row stride 25 doubles:
// conflict-free).  rows == null: observations q0 .. q0+nrows-1 (contiguous); else observation rows[q0 + i].
// Measured on Venice (bench.py kernel classes, ms per launch, plain -> staged): right-hand side pass (k_cam_blocks<2>)
// 0.68 -> 0.47; back-substitution + model value in one staged pass 0.48 + 0.37 -> 0.76; but the 54-accumulator Hcc pass
// 0.32 -> 0.50 and the point-block pass 0.27 -> 0.33 got SLOWER and keep their plain loads: the vector L1 already merges
// the 24 strided loads of a wave, and the PMC "over-fetch" of those kernels (FETCH_SIZE x 2) overstates 8-byte loads.
constexpr int JLD = 25;
constexpr int ST_WAVE_ELEMS = 64 * JLD;
typedef double d2n __attribute__((ext_vector_type(2)));

__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the two halves of stage_rows: the 12 requests of a batch (in registers until they are committed), and their way into the
// wave's LDS slot.  Issued one batch ahead they overlap the work on the current batch (k_wtv).
__device__ inline void stage_issue(const double *__restrict__ J, const int *__restrict__ rows, int64_t q0, int nrows, d2n v[12]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int ps = 0; ps < 12; ps++) {  // piece f = 64 ps + lane of the 768 16-byte pieces: row f / 12, piece f % 12
    const int f = ps * 64 + lane;
    int r = f / 12;
    const int pc = f - 12 * r;
    r = r < nrows ? r : nrows - 1;  // clamped, not predicated: a load under a per-lane condition serialises
    const int64_t o = rows ? (int64_t)rows[q0 + r] : q0 + r;
    v[ps] = *reinterpret_cast<const d2n *>(J + 24 * o + 2 * pc);
  }
}
__device__ inline void stage_commit(const d2n v[12], double *slot) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int ps = 0; ps < 12; ps++) {
    const int f = ps * 64 + lane;
    const int r = f / 12, pc = f - 12 * r;
    slot[r * JLD + 2 * pc] = v[ps].x;
    slot[r * JLD + 2 * pc + 1] = v[ps].y;
  }
  wave_lds_sync();
}
__device__ inline void stage_rows(const double *__restrict__ J, const int *__restrict__ rows, int64_t q0, int nrows,
                                  double *slot) {
  d2n v[12];
  stage_issue(J, rows, q0, nrows, v);
  stage_commit(v, slot);
}
// the same with the 12 row numbers of a lane's pieces already at hand (stage_row_ids, requested one batch ahead: the chain
// index -> row then costs one memory round trip per batch instead of two)
__device__ inline void stage_row_ids(const int *__restrict__ rows, int64_t q0, int nrows, int ro[12]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int ps = 0; ps < 12; ps++) {
    int r = (ps * 64 + lane) / 12;
    r = r < nrows ? r : nrows - 1;
    ro[ps] = rows[q0 + r];
  }
}
__device__ inline void stage_rows_by_id(const double *__restrict__ J, const int ro[12], double *slot) {
  const int lane = threadIdx.x & 63;
  d2n v[12];
#pragma unroll
  for (int ps = 0; ps < 12; ps++) {
    const int f = ps * 64 + lane, pc = f - 12 * (f / 12);
    v[ps] = *reinterpret_cast<const d2n *>(J + 24 * (int64_t)ro[ps] + 2 * pc);
  }
  stage_commit(v, slot);
}

// camera side with staged rows: same work item, same summation order as k_cam_blocks (thread t of camera c takes list
// positions t, t + 256, ...), hence the same bits
template <int MODE>
__global__ __launch_bounds__(BLK) void k_cam_blocks_st(const int *__restrict__ cam_ptr, const int *__restrict__ cam_obs,
                                                        const int *__restrict__ pnt0, const double *__restrict__ J,
                                                        const double *__restrict__ r, const double *__restrict__ u,
                                                        double *__restrict__ Hcc, double *__restrict__ out9, double lam = 0.0,
                                                        const int *__restrict__ cam_pnt = nullptr,
                                                        const int *__restrict__ opos = nullptr) {
  constexpr int NACC = (MODE == 0) ? 54 : 9;
  __shared__ double slots[(BLK / 64) * ST_WAVE_ELEMS];
  __shared__ double red[BLK / 64][NACC];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *slot = slots + wv * ST_WAVE_ELEMS;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  const int qend = cam_ptr[c + 1];
  int ro[12], ron[12];  // row numbers of this lane's pieces: current batch, next batch
  if (cam_ptr[c] + wv * 64 < qend) {
    const int q0 = cam_ptr[c] + wv * 64;
    stage_row_ids(cam_obs, q0, qend - q0 < 64 ? qend - q0 : 64, ro);
  }
  for (int q0 = cam_ptr[c] + wv * 64; q0 < qend; q0 += BLK) {  // wave-uniform
    const int nrows = qend - q0 < 64 ? qend - q0 : 64;
    // MODE 2 / 3: the point-side vector of this lane's observation is requested BEFORE the rows are staged, so that the
    // chain index -> point index -> u overlaps the chain index -> row instead of following it (cam_pnt = the point index in
    // camera order, one indirection less): 117.9 -> 107.3 ms over the 288 products of a Venice PCG run.  (Requesting the
    // rows themselves one batch ahead, as k_wtv does, made this kernel slower: 124 ms; the row NUMBERS are requested ahead.)
    double up3[3] = {0, 0, 0};
    if ((MODE == 2 || MODE == 3) && lane < nrows) {
      const int64_t pi = cam_pnt ? cam_pnt[q0 + lane] : pnt0[cam_obs[q0 + lane]];
      up3[0] = u[3 * pi];
      up3[1] = u[3 * pi + 1];
      up3[2] = u[3 * pi + 2];
    }
    const bool more = q0 + BLK < qend;
    if (more) stage_row_ids(cam_obs, q0 + BLK, qend - (q0 + BLK) < 64 ? qend - (q0 + BLK) : 64, ron);  // in flight behind the rows
    stage_rows_by_id(J, ro, slot);
    if (more) {
#pragma unroll
      for (int ps = 0; ps < 12; ps++) ro[ps] = ron[ps];
    }
    if (lane < nrows) {
      const int64_t o = cam_obs[q0 + lane];
      const double *Jo = slot + lane * JLD;
      double b0[9], b1[9];
#pragma unroll
      for (int i = 0; i < 9; i++) {
        b0[i] = Jo[3 + i];
        b1[i] = Jo[15 + i];
      }
      double w0, w1;
      if (MODE == 2) {
        const double *up = up3;
        w0 = (Jo[0] * up[0] + Jo[1] * up[1] + Jo[2] * up[2]) - r[2 * o];
        w1 = (Jo[12] * up[0] + Jo[13] * up[1] + Jo[14] * up[2]) - r[2 * o + 1];
      } else if (MODE == 3) {
        const double *up = up3;
        w0 = Jo[0] * up[0] + Jo[1] * up[1] + Jo[2] * up[2];
        w1 = Jo[12] * up[0] + Jo[13] * up[1] + Jo[14] * up[2];
      } else {
        w0 = r[2 * o];
        w1 = r[2 * o + 1];
      }
      if (MODE == 0) {
        int idx = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) acc[idx++] += b0[i] * b0[j] + b1[i] * b1[j];
#pragma unroll
        for (int i = 0; i < 9; i++) acc[45 + i] += b0[i] * w0 + b1[i] * w1;
      } else {
#pragma unroll
        for (int i = 0; i < 9; i++) acc[i] += b0[i] * w0 + b1[i] * w1;
      }
    }
    wave_lds_sync();  // the slot is rewritten by the next batch
  }
#pragma unroll
  for (int i = 0; i < NACC; i++) {
    double v = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    if (MODE == 0) {
      if (threadIdx.x < 45) Hcc[45 * (int64_t)c + threadIdx.x] = v;
      else out9[9 * (int64_t)c + threadIdx.x - 45] = v;
    } else if (MODE == 3) {  // PCG product: + Hcc_c x_c + lam x_c, x handed over in `r` (Hcc packed lower, read only)
      const int i = threadIdx.x;
      const double *x = r + 9 * (int64_t)c;
      const double *H = Hcc + 45 * (int64_t)c;
      double s = 0;
#pragma unroll
      for (int j = 0; j < 9; j++) s += H[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * x[j];
      out9[9 * (int64_t)c + i] = (s + v) + lam * x[i];
    } else {
      out9[9 * (int64_t)((MODE == 2 && opos) ? opos[c] : c) + threadIdx.x] = v;
    }
  }
}

// ---- per point: U^-1 = (Hpp + lambda I)^-1 (6, symmetric) and u = U^-1 gp -------------------------------------
// lam_dev (here and below): when non-null the damping is lambda * lam_dev[0] -- a launch recorded in a hipGraph keeps
// its arguments, so the replayed LM iteration reads the current damping from device memory.
// damp (Float16 path only): a per-variable damping vector in the layout of x replaces lambda I (see k_f16_cols).
__global__ __launch_bounds__(BLK) void k_schur_prep(int64_t npnts, double lambda, const double *__restrict__ lam_dev,
                                                     const double *__restrict__ Hpp, const double *__restrict__ gp,
                                                     double *__restrict__ Uinv, double *__restrict__ u,
                                                     const double *__restrict__ damp, double *__restrict__ lam_copy,
                                                     double *__restrict__ zero, int64_t nzero) {
  int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x;
  // first kernel of a step: two chores ride along that were launches of their own (recorded sequences pay ~5 us per
  // node): the damping, read here from pinned host memory, is left in device memory for the kernels that follow, and the
  // right-hand side of the camera system is cleared for the sums of k_schur_rhs
  if (lam_copy && p == 0) lam_copy[0] = lam_dev[0];
  for (int64_t i = p; i < nzero; i += (int64_t)gridDim.x * BLK) zero[i] = 0.0;
  if (p >= npnts) return;
  if (lam_dev) lambda *= lam_dev[0];
  const double *h = Hpp + 6 * p;
  const double l0 = damp ? damp[3 * p] : lambda, l1 = damp ? damp[3 * p + 1] : lambda, l2 = damp ? damp[3 * p + 2] : lambda;
  double a = h[0] + l0, b = h[1], c = h[2], d = h[3] + l1, e = h[4], f = h[5] + l2;
  // cofactors of the symmetric 3x3 [[a b c],[b d e],[c e f]]
  double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
  double det = a * c00 + b * c01 + c * c02;
  double id = 1.0 / det;
  double i00 = c00 * id, i01 = c01 * id, i02 = c02 * id, i11 = c11 * id, i12 = c12 * id, i22 = c22 * id;
  double *o = Uinv + 6 * p;
  o[0] = i00;
  o[1] = i01;
  o[2] = i02;
  o[3] = i11;
  o[4] = i12;
  o[5] = i22;
  const double *g = gp + 3 * p;
  u[3 * p + 0] = i00 * g[0] + i01 * g[1] + i02 * g[2];
  u[3 * p + 1] = i01 * g[0] + i11 * g[1] + i12 * g[2];
  u[3 * p + 2] = i02 * g[0] + i12 * g[1] + i22 * g[2];
}

// ---- Schur blocks: one wave per (camera_a >= camera_b) key ---------------------------------------------------------
// S[ca, cb] = [ca == cb] (Hcc[ca] + lambda I) - sum_tasks B_a' Q_ab B_b,   Q_ab = A_a U_p^-1 A_b'.
// Per task the wave loads J_a (24), J_b (24) and U_p^-1 (6) with one coalesced instruction into its LDS slot,
// lane l then owns elements l and 64+l (< 81) of the 9x9 block.  Only elements with global row >= col are written
// (lower triangle, packed NB x NB tiles).
__device__ inline void s_store(double *S, const int64_t *__restrict__ co, int64_t gr, int64_t gc, double v) {
  int64_t ti = gr / NB, tj = gc / NB;
  if (co[tj] < 0) return;  // (chunked assembly of a distributed run: this tile column belongs to another chunk)
  const int64_t tt = tix(co, ti, tj);
  if (tt < 0) return;  // (compressed storage: cannot happen for a key of the list the pattern was built from)
  S[(tt * NB + (gr - ti * NB)) * NB + (gc - tj * NB)] = v;
}

// Y_b = U_p^-1 A_b'  (3x2, row-major) per observation: the point-side half of Q_ab = A_a Y_b
__global__ __launch_bounds__(BLK) void k_obs_y(int64_t nobs, const int *__restrict__ pnt0, const double *__restrict__ J,
                                                const double *__restrict__ Uinv, double *__restrict__ Y) {
  int64_t o = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (o >= nobs) return;
  const double *Jo = J + 24 * o, *U = Uinv + 6 * (int64_t)pnt0[o];
  const double a00 = Jo[0], a01 = Jo[1], a02 = Jo[2], a10 = Jo[12], a11 = Jo[13], a12 = Jo[14];
  double *y = Y + 6 * o;
  y[0] = U[0] * a00 + U[1] * a01 + U[2] * a02;
  y[1] = U[0] * a10 + U[1] * a11 + U[2] * a12;
  y[2] = U[1] * a00 + U[3] * a01 + U[4] * a02;
  y[3] = U[1] * a10 + U[3] * a11 + U[4] * a12;
  y[4] = U[2] * a00 + U[4] * a01 + U[5] * a02;
  y[5] = U[2] * a10 + U[4] * a11 + U[5] * a12;
}

// (Round 3, measured and NOT adopted: a packed per-observation record [B_b (18) | Y_b (6)], 192 aligned bytes, so that a
// task is two contiguous reads -- exactly its 384 algorithmic bytes instead of the ~660 the PMC counters show.  rocprofv3 on
// Venice: k_schur_blocks 2.66 -> 2.58 ms, but k_obs_y, which then writes 192 instead of 48 bytes per observation, 0.28 ->
// 0.45 ms and k_schur_chunks 0.36 -> 0.41 ms: a net loss.  The kernel is not bound by the bytes of its gathers: at 80 VGPRs
// six waves per SIMD keep 12 task pairs in flight, and the request latency of that many dependent gathers is what one sees.)
// One wave per key.  The sum over the key's tasks of B_a' (Q_ab B_b) is a (9 x 2t)(2t x 9) product: two tasks fill
// the four k-slots of one v_mfma_f64_16x16x4_f64 (slot = (task, image row alpha)); the A operand is B_a[alpha][i], the
// B operand G[alpha][j] = Q_ab[alpha][:] B_b[:][j] with Q_ab = A_a Y_b.  Per task one coalesced load of
// J_a (24), the camera rows of J_b (18) and Y_b (6) is staged in the wave's LDS slot.
typedef double d4s __attribute__((ext_vector_type(4)));
#ifndef BA_SCHUR_PF
#define BA_SCHUR_PF 4  // task pairs in flight per wave (ring of LDS slots, see schur_accumulate)
#endif
constexpr int SCHUR_PF = BA_SCHUR_PF;
constexpr int SCHUR_SLOT = 128;                       // doubles per ring slot: 64 lanes x 16 bytes
constexpr int SCHUR_RING = SCHUR_PF * SCHUR_SLOT;     // doubles per wave
// sum over the tasks [t_begin, t_end) of B_a' (Q_ab B_b): lane (fr < 9, i = fk + 4 g < 9) gets element (i, j = fr) in acc[g]
//
// A task's record -- J_a (24 doubles), the camera columns of J_b (2 x 9) and Y_b (6) -- is gathered as 25 aligned 16-byte
// pieces: 12 of J_a, 5 + 5 of J_b (doubles 2..11 and 14..23: one point column rides along in each half so that the pieces
// stay aligned), 3 of Y_b.  ONE load instruction fetches a PAIR of tasks (lanes 0..24 the first, 32..56 the second) and
// delivers it straight into the wave's LDS ring (global_load_lds_dwordx4: no staging registers, no ds_write), SCHUR_PF
// pairs ahead; the wave waits with vmcnt only.  Record in its slot (doubles, task tl at 64 tl): J_a at 0..23, J_b[3 + i] at
// 25 + i, J_b[15 + i] at 35 + i, Y_b at 44..49.
// (Round 3's form staged single elements through a rotating set of prefetch REGISTERS (pf[q] = pf[q + 1]); a move of a
// register whose load is in flight makes the compiler wait for that load, so the generated loop began with s_waitcnt
// vmcnt(0): every trip waited for the loads issued one trip earlier -- a whole memory round trip, ~4 700 cycles per task
// pair and wave in the kernel trace -- whatever the depth of the ring; which is why depths 2, 3, 4 had measured the same.)
// PRE: the observation indices of the first 64 tasks come in registers (pre_oa / pre_ob: lane l holds task t_begin + l) --
// k_schur_blocks loads them one key ahead, so that a key does not start with a memory round trip for its own task list.
template <bool PRE = false>
__device__ __forceinline__ d4s schur_accumulate(int t_begin, int t_end, const int *__restrict__ task_a, const int *__restrict__ task_b,
                                                const double *__restrict__ J, const double *__restrict__ Y, double *ring,
                                                int pre_oa = 0, int pre_ob = 0) {
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fk = lane >> 4, tl = fk >> 1, al = fk & 1;
  d4s acc = {0, 0, 0, 0};
  // (the task range is wave-uniform but arrives in vector registers: made scalar, or every trip count below is a lane mask)
  t_begin = __builtin_amdgcn_readfirstlane(t_begin);
  t_end = __builtin_amdgcn_readfirstlane(t_end);
  // this lane's piece as ONE address expression, J + 8 (mul * observation + add), the Y block reached through its distance
  // from J: a fetch is two v_readlane, a select, a multiply-add and the load -- no branch.  Idle lanes re-read piece 0.
  const int pc = (lane & 31) < 25 ? (lane & 31) : 0;  // piece of the record
  const bool second = lane >= 32;                     // ... of the pair's second task
  const bool from_a = pc < 12;
  const int64_t ydelta = (int64_t)((uintptr_t)Y - (uintptr_t)J) >> 3;  // (signed: Y may lie below J)
  const int64_t mul = pc < 22 ? 24 : 6;
  const int64_t add = pc < 12 ? 2 * pc : (pc < 17 ? 2 + 2 * (pc - 12) : (pc < 22 ? 14 + 2 * (pc - 17) : ydelta + 2 * (pc - 22)));
  for (int c0t = t_begin; c0t < t_end; c0t += 64) {
    const int nin = (t_end - c0t) < 64 ? (t_end - c0t) : 64;
    int my_oa = 0, my_ob = 0;
    if (PRE && c0t == t_begin) {
      my_oa = pre_oa;
      my_ob = pre_ob;
    } else if (lane < nin) {
      my_oa = task_a[c0t + lane];
      my_ob = task_b[c0t + lane];
    }
    auto issue = [&](int pr) {  // tasks 2 pr, 2 pr + 1 of this chunk (clamped to its last) -> slot pr % SCHUR_PF
      int t0 = 2 * pr, t1 = 2 * pr + 1;
      t0 = t0 < nin ? t0 : nin - 1;
      t1 = t1 < nin ? t1 : nin - 1;
      const int oa0 = __builtin_amdgcn_readlane(my_oa, t0), ob0 = __builtin_amdgcn_readlane(my_ob, t0);
      const int oa1 = __builtin_amdgcn_readlane(my_oa, t1), ob1 = __builtin_amdgcn_readlane(my_ob, t1);
      const int64_t o = second ? (from_a ? oa1 : ob1) : (from_a ? oa0 : ob0);
      __builtin_amdgcn_global_load_lds(J + (mul * o + add), ring + (pr % SCHUR_PF) * SCHUR_SLOT, 16, 0, 0);
    };
    const int npairs = (nin + 1) >> 1;
#pragma unroll
    for (int q = 0; q < SCHUR_PF; q++) issue(q);
    for (int pr0 = 0; pr0 < npairs; pr0 += SCHUR_PF) {
#pragma unroll
      for (int q = 0; q < SCHUR_PF; q++) {
        const int pr = pr0 + q;
        if (pr >= npairs) break;  // wave-uniform
        // the oldest load of the ring has landed when at most SCHUR_PF - 1 are outstanding (vmcnt retires in order)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SCHUR_PF - 1) : "memory");
        const double *sg = ring + q * SCHUR_SLOT + 64 * tl;
        const bool on = (fr < 9) && (2 * pr + tl < nin);
        const int fi = fr < 9 ? fr : 0;
        // (every operand is read unconditionally -- the addresses are valid in every lane -- and masked afterwards: as
        // `on ? sg[..] : 0` the reads sat in branches of their own, three LDS round trips in sequence per trip)
        const double a0 = sg[12 * al], a1 = sg[12 * al + 1], a2 = sg[12 * al + 2];
        const double av = sg[12 * al + 3 + fi], b0 = sg[25 + fi], b1 = sg[35 + fi];
        const double q0 = a0 * sg[44] + a1 * sg[46] + a2 * sg[48];
        const double q1 = a0 * sg[45] + a1 * sg[47] + a2 * sg[49];
        const double aop = on ? av : 0.0;
        const double bop = on ? q0 * b0 + q1 * b1 : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
        // the slot is free once its operands are in registers (the matrix instruction above has them): refill it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(pr + SCHUR_PF);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped loads of the last trips: the ring is reused by the next chunk
  }
  return acc;
}

// block (ca, cb) of S from the task sum `sum` (element (i, j) per lane as above; i >= 9 or fr >= 9: ignored)
// ca, cb are block rows of S; cam_of (null: the identity) gives the camera that sits there (fill-reducing camera ordering)
__device__ inline void schur_store_block(double *S, const int64_t *__restrict__ co, int ca, int cb, const double *__restrict__ Hcc,
                                         double lambda, const double *__restrict__ damp_c, int i, int j, double sum,
                                         const int *__restrict__ cam_of) {
  double v = -sum;
  if (ca == cb) {
    int hi = i > j ? i : j, lo = i > j ? j : i;
    const int64_t cam = cam_of ? cam_of[ca] : ca;
    v += Hcc[45 * cam + hi * (hi + 1) / 2 + lo] + (i == j ? (damp_c ? damp_c[9 * cam + i] : lambda) : 0.0);
  }
  const int64_t r0 = 9 * (int64_t)ca, c0 = 9 * (int64_t)cb;
  if (r0 + i >= c0 + j) s_store(S, co, r0 + i, c0 + j, v);
}

// one wave per key; keys longer than split_above tasks (> 0) are left to k_schur_chunks / k_schur_combine
__global__ __launch_bounds__(BLK, 6) void k_schur_blocks(int64_t nkeys, const int *__restrict__ key_ptr,
                                                       const int *__restrict__ key_ca, const int *__restrict__ key_cb,
                                                       const int *__restrict__ task_a, const int *__restrict__ task_b,
                                                       const double *__restrict__ J, const double *__restrict__ Y,
                                                       const double *__restrict__ Hcc, double lambda,
                                                       const double *__restrict__ lam_dev, double *__restrict__ S,
                                                       const int64_t *__restrict__ co, const double *__restrict__ damp_c,
                                                       int split_above, const int *__restrict__ klist = nullptr,
                                                       const int *__restrict__ cam_of = nullptr) {
  // klist (per-rank ownership of S): the keys whose blocks touch the tile columns of the chunk being assembled; nkeys is
  // then the length of that list
  __shared__ __attribute__((aligned(16))) double ring[BLK / 64][SCHUR_RING];
  if (lam_dev) lambda *= lam_dev[0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  // grid-stride over the keys: the AQL grid size is a 32-bit count of work-items, and Final-13682 has 93.6 M keys
  // (x 64 lanes > 2^32): a one-wave-per-key launch silently dropped the tail (zero pivots at the first lost diagonal).
  // (Tried: giving each XCD a contiguous eighth of the (camera_a, camera_b)-sorted keys so that a camera's own rows are
  // re-read from that XCD's L2 -- 4.3 -> 5.1 ms on Venice in round 1's kernel, 3.73 -> 3.67 ms in this one: the L2-miss traffic
  // (9.5 GB per launch against 3.4 GB if every row were fetched once per XCD that needs it) is served by the MALL and does not
  // bound the kernel; the interleaved order stays.)
  // (the wave's key index is uniform: told so, the compiler reads the key tables through the scalar cache into scalar registers)
  const int64_t stride = (int64_t)gridDim.x * (BLK / 64), first = (int64_t)blockIdx.x * (BLK / 64) + __builtin_amdgcn_readfirstlane(wv);
  if (klist) {  // (chunked assembly of a distributed run: the keys come through a list -- one more dependent load; plain loop)
    for (int64_t kq = first; kq < nkeys; kq += stride) {
      const int64_t key = klist[kq];
      const int t_begin = key_ptr[key], t_end = key_ptr[key + 1];
      if (split_above > 0 && t_end - t_begin > split_above) continue;
      const d4s acc = schur_accumulate(t_begin, t_end, task_a, task_b, J, Y, ring[wv]);
      const int ca = key_ca[key], cb = key_cb[key];
      if (fr < 9) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int i = fk + 4 * g;
          if (i < 9) schur_store_block(S, co, ca, cb, Hcc, lambda, damp_c, i, fr, acc[g], cam_of);
        }
      }
    }
    return;
  }
  // A key is ~10 tasks on Venice: five trips of the ring between three dependent memory round trips (its range in key_ptr,
  // its task list, its first records), and the counters show the waves parked on s_waitcnt 70 % of their cycles.  So the
  // wave runs one key AHEAD with its lists: while key i is summed the task list of key i + 1 is in flight (and the range of
  // key i + 2, which that load needs), and a key starts with its record loads.  (A load issued BEFORE the ring's loads is
  // older than they are: the hand-counted vmcnt waits of the ring stay valid.)
  // (what is wave-uniform moves to scalar registers once it has arrived: five waves per SIMD with everything in vector
  // registers, six as before with this)
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto list_of = [&](int tb, int te, int &oa, int &ob) {
    oa = ob = 0;
    if (lane < te - tb) {  // (at most the first 64 tasks: what the first trip of schur_accumulate needs)
      oa = task_a[tb + lane];
      ob = task_b[tb + lane];
    }
  };
  int tb0 = 0, te0 = 0, ca0 = 0, cb0 = 0, tb1 = 0, te1 = 0, oa0, ob0;
  if (first < nkeys) {
    tb0 = uni(key_ptr[first]);
    te0 = uni(key_ptr[first + 1]);
    ca0 = uni(key_ca[first]);
    cb0 = uni(key_cb[first]);
  }
  if (first + stride < nkeys) {
    tb1 = uni(key_ptr[first + stride]);
    te1 = uni(key_ptr[first + stride + 1]);
  }
  list_of(tb0, te0, oa0, ob0);
  for (int64_t kq = first; kq < nkeys; kq += stride) {
    int tb2 = 0, te2 = 0, ca1 = 0, cb1 = 0, oa1, ob1;  // in flight while key kq is summed
    if (kq + 2 * stride < nkeys) {
      tb2 = key_ptr[kq + 2 * stride];
      te2 = key_ptr[kq + 2 * stride + 1];
    }
    if (kq + stride < nkeys) {
      ca1 = key_ca[kq + stride];
      cb1 = key_cb[kq + stride];
    }
    list_of(tb1, te1, oa1, ob1);
    if (!(split_above > 0 && te0 - tb0 > split_above)) {
      const d4s acc = schur_accumulate<true>(tb0, te0, task_a, task_b, J, Y, ring[wv], oa0, ob0);
      if (fr < 9) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int i = fk + 4 * g;
          if (i < 9) schur_store_block(S, co, ca0, cb0, Hcc, lambda, damp_c, i, fr, acc[g], cam_of);
        }
      }
    }
    tb0 = tb1; te0 = te1; oa0 = oa1; ob0 = ob1;
    ca0 = uni(ca1); cb0 = uni(cb1); tb1 = uni(tb2); te1 = uni(te2);
  }
}

// one wave per chunk of a split key: partial[chunk][i * 9 + j]
__global__ __launch_bounds__(BLK) void k_schur_chunks(int64_t nchunks, const int *__restrict__ chunk_t0, const int *__restrict__ chunk_t1,
                                                       const int *__restrict__ task_a, const int *__restrict__ task_b,
                                                       const double *__restrict__ J, const double *__restrict__ Y,
                                                       double *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) double ring[BLK / 64][SCHUR_RING];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  for (int64_t c = (int64_t)blockIdx.x * (BLK / 64) + wv; c < nchunks; c += (int64_t)gridDim.x * (BLK / 64)) {
    const d4s acc = schur_accumulate(chunk_t0[c], chunk_t1[c], task_a, task_b, J, Y, ring[wv]);
    if (fr < 9) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int i = fk + 4 * g;
        if (i < 9) partial[81 * c + 9 * i + fr] = acc[g];
      }
    }
  }
}

// one wave per split key: the chunk partials are added in chunk order (fixed) and the block is stored
__global__ __launch_bounds__(BLK) void k_schur_combine(int64_t nsplit, const int *__restrict__ skey, const int *__restrict__ skey_c0,
                                                        const int *__restrict__ key_ca, const int *__restrict__ key_cb,
                                                        const double *__restrict__ partial, const double *__restrict__ Hcc,
                                                        double lambda, const double *__restrict__ lam_dev, double *__restrict__ S,
                                                        const int64_t *__restrict__ co, const double *__restrict__ damp_c,
                                                        const int *__restrict__ slist = nullptr,
                                                        const int *__restrict__ cam_of = nullptr) {
  if (lam_dev) lambda *= lam_dev[0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t sq = (int64_t)blockIdx.x * (BLK / 64) + wv; sq < nsplit; sq += (int64_t)gridDim.x * (BLK / 64)) {
    const int64_t q = slist ? slist[sq] : sq;  // slist: the split keys of the chunk being assembled
    const int key = skey[q], c0 = skey_c0[q], c1 = skey_c0[q + 1];
    const int ca = key_ca[key], cb = key_cb[key];
    for (int e = lane; e < 81; e += 64) {
      double sum = 0;
      for (int c = c0; c < c1; c++) sum += partial[81 * (int64_t)c + e];
      schur_store_block(S, co, ca, cb, Hcc, lambda, damp_c, e / 9, e % 9, sum, cam_of);
    }
  }
}

// Column scaling of the reduced camera system (normalize = :J / :A, reference: src/lma_aux.jl:102-154): the reference
// scales the columns of J by their 2-norms (:J) or those of [J; sqrt(lambda) I] (:A) before factorising.  The point
// columns are eliminated in closed 3x3 form here, so only the camera columns matter: d_j = sqrt(diag(Hcc)_j [+ lambda]),
// S <- D^-1 S D^-1, rhs <- D^-1 rhs, and afterwards dc <- D^-1 dc'.  Same step in exact arithmetic, better conditioned.
// hdiag: diag(J'J) of the camera columns, summed over ALL ranks (k_hcc_diag + the gc all-reduce): the column norms are
// those of the whole Jacobian, not of a rank's shard.
__global__ __launch_bounds__(BLK) void k_cam_scale(int64_t ncams, const double *__restrict__ hdiag, double add,
                                                    const double *__restrict__ lam_dev, double *__restrict__ dsc,
                                                    const int *__restrict__ pos) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= 9 * ncams) return;
  if (lam_dev) add *= lam_dev[0];
  double v = sqrt(hdiag[i] + add);
  // (pos: the block row of S a camera sits at under a fill-reducing camera ordering; dsc is indexed like S)
  const int64_t o = pos ? 9 * (int64_t)pos[i / 9] + i % 9 : i;
  dsc[o] = (v != 0.0) ? v : 1.0;  // col_norms[j] == 0 columns are left alone (lma_aux.jl:120,148)
}

// this rank's diag(Hcc) (packed lower 9x9 per camera) -> the 9*ncams vector that is all-reduced with gc
__global__ __launch_bounds__(BLK) void k_hcc_diag(int64_t ncams, const double *__restrict__ Hcc, double *__restrict__ hdiag) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= 9 * ncams) return;
  int64_t c = i / 9;
  int j = (int)(i - 9 * c);
  hdiag[i] = Hcc[45 * c + j * (j + 1) / 2 + j];
}

// one workgroup per stored tile: S_ij /= d_i d_j   (padding rows/columns have d = 1)
__global__ __launch_bounds__(BLK) void k_scale_S(int64_t n, int64_t nt, const double *__restrict__ dsc, double *__restrict__ S,
                                                  const int64_t *__restrict__ co) {
  const int64_t t = blockIdx.x;  // enumerates the lower tile pairs (ti >= tj) row by row; the storage order is co's
  int64_t ti = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
  while (ti * (ti + 1) / 2 > t) ti--;
  const int64_t tj = t - ti * (ti + 1) / 2;
  const int64_t tt = tix(co, ti, tj);
  if (tt < 0) return;  // (compressed block-sparse storage: not in the pattern)
  double *T = S + tt * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += BLK) {
    const int64_t r = ti * NB + (e >> 7), c = tj * NB + (e & (NB - 1));
    const double dr = r < n ? dsc[r] : 1.0, dc = c < n ? dsc[c] : 1.0;
    T[e] /= dr * dc;
  }
}

// unit diagonal on the padding rows n..npad-1 so that the padded matrix stays factorisable
__global__ void k_pad_diag(int64_t n, int64_t npad, double *S, const int64_t *co) {
  int64_t i = n + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) s_store(S, co, i, i, 1.0);
}

// ---- facto_type = Float16 (src/lm.jl:165-169, src/lma_aux.jl:30-52,90-95) ------------------------------------------------
// The reference scales every column j of the upper triangle of K = [[I J];[J' -lambda I]] by its 2-norm D_j, multiplies by
// mu = 0.1 * 65500, rounds to Float16 and factors THAT matrix in Float16; the right-hand side is [-r; 0] rounded to Float16
// and the solution's variable block is taken as the step without any back-scaling.  D_j = 1 on the residual rows and
// sqrt(|J_j|^2 + lambda^2) on the variable columns, so in exact arithmetic the step is
//     (Jh' Jh + Lh) y = -Jh' r / mu,   Jh = J D^-1,  Lh = diag(lambda / D_j),   delta = y,   delta_r = -r / mu - Jh y.
// The device keeps what the reference's inputs lose -- Jh, Lh and r go through Float16 -- and then eliminates and factors in
// Float32 (the reduced camera system), instead of carrying Float16 through ~10^5 pivots.
__device__ inline double round_f16(double v) { return (double)__half2float(__float2half_rn((float)v)); }

// column norms and damping of the variable columns: dcol[j] = sqrt(jn2[j] + lambda^2), damp[j] = f16(mu lambda / dcol[j]) / mu
__global__ __launch_bounds__(BLK) void k_f16_cols(int64_t nvar, const double *__restrict__ jn2, double lambda, double mu,
                                                   double *__restrict__ dcol, double *__restrict__ damp) {
  int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (j >= nvar) return;
  const double d = sqrt(jn2[j] + lambda * lambda);
  dcol[j] = d;
  damp[j] = round_f16(mu * (lambda / d)) / mu;
}

// |J_j|^2 of every column from the diagonals of the normal-equation blocks (layout of x)
__global__ __launch_bounds__(BLK) void k_col_sq(int64_t npnts, int64_t ncams, const double *__restrict__ Hpp,
                                                 const double *__restrict__ hdiag, double *__restrict__ jn2) {
  int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (j < 3 * npnts) {
    const int c = (int)(j % 3);
    jn2[j] = Hpp[6 * (j / 3) + (c == 0 ? 0 : (c == 1 ? 3 : 5))];
  } else if (j < 3 * npnts + 9 * ncams) {
    jn2[j] = hdiag[j - 3 * npnts];
  }
}

// Jq = f16(mu J / D) / mu, rq = f16(r): one lane per observation
__global__ __launch_bounds__(BLK) void k_f16_quantize(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                       const int *__restrict__ pnt0, const double *__restrict__ J,
                                                       const double *__restrict__ r, const double *__restrict__ dcol, double mu,
                                                       double *__restrict__ Jq, double *__restrict__ rq) {
  int64_t o = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (o >= nobs) return;
  const double *dp = dcol + 3 * (int64_t)pnt0[o], *dc = dcol + 3 * npnts + 9 * (int64_t)cam0[o];
  const double *Jo = J + 24 * o;
  double *Qo = Jq + 24 * o;
#pragma unroll
  for (int a = 0; a < 2; a++) {
#pragma unroll
    for (int c = 0; c < 3; c++) Qo[12 * a + c] = round_f16(mu * (Jo[12 * a + c] / dp[c])) / mu;
#pragma unroll
    for (int c = 0; c < 9; c++) Qo[12 * a + 3 + c] = round_f16(mu * (Jo[12 * a + 3 + c] / dc[c])) / mu;
  }
  rq[2 * o] = round_f16(r[2 * o]);
  rq[2 * o + 1] = round_f16(r[2 * o + 1]);
}

// ---- back-substitution of the points: dp = -(u_p + U^-1 sum_a A_a' (B_a dc[c_a])) ----------------------------------
__global__ __launch_bounds__(BLK) void k_backsub(int64_t npnts, const int *__restrict__ pt_ptr,
                                                  const int *__restrict__ pt_obs, const int *__restrict__ cam0,
                                                  const double *__restrict__ J, const double *__restrict__ Uinv,
                                                  const double *__restrict__ u, const double *__restrict__ dc,
                                                  double *__restrict__ dp) {
  int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= npnts) return;
  double w[3] = {0, 0, 0};
  for (int q = pt_ptr[p]; q < pt_ptr[p + 1]; q++) {
    int64_t o = pt_obs[q];
    const double *Jo = J + 24 * o;
    const double *d = dc + 9 * (int64_t)cam0[o];
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      s0 += Jo[3 + i] * d[i];
      s1 += Jo[15 + i] * d[i];
    }
    w[0] += Jo[0] * s0 + Jo[12] * s1;
    w[1] += Jo[1] * s0 + Jo[13] * s1;
    w[2] += Jo[2] * s0 + Jo[14] * s1;
  }
  const double *U = Uinv + 6 * p;
  dp[3 * p + 0] = -(u[3 * p + 0] + (U[0] * w[0] + U[1] * w[1] + U[2] * w[2]));
  dp[3 * p + 1] = -(u[3 * p + 1] + (U[1] * w[0] + U[3] * w[1] + U[4] * w[2]));
  dp[3 * p + 2] = -(u[3 * p + 2] + (U[2] * w[0] + U[4] * w[1] + U[5] * w[2]));
}

// ---- model value: per-block partial sums of |J delta + r|^2 (delta in the layout of x) ------------------------------
__global__ __launch_bounds__(BLK) void k_model_sq(int64_t nobs, int64_t npnts, const int *__restrict__ cam0,
                                                   const int *__restrict__ pnt0, const double *__restrict__ J,
                                                   const double *__restrict__ r, const double *__restrict__ delta,
                                                   double cr, double *__restrict__ partial) {
  __shared__ double red[BLK / 64];
  double acc = 0;
  for (int64_t o = (int64_t)blockIdx.x * BLK + threadIdx.x; o < nobs; o += (int64_t)gridDim.x * BLK) {
    const double *Jo = J + 24 * o;
    const double *dp = delta + 3 * (int64_t)pnt0[o];
    const double *dcm = delta + 3 * npnts + 9 * (int64_t)cam0[o];
    double s0 = cr * r[2 * o], s1 = cr * r[2 * o + 1];  // cr == 1 except inside the line search with delta_d != 2
#pragma unroll
    for (int i = 0; i < 3; i++) {
      s0 += Jo[i] * dp[i];
      s1 += Jo[12 + i] * dp[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      s0 += Jo[3 + i] * dcm[i];
      s1 += Jo[15 + i] * dcm[i];
    }
    acc += s0 * s0 + s1 * s1;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// sum of squares of a vector -> per-block partials (fixed grid => fixed summation tree)
__global__ __launch_bounds__(BLK) void k_sumsq(int64_t n, const double *__restrict__ v, double *__restrict__ partial) {
  __shared__ double red[BLK / 64];
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) acc += v[i] * v[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// final stage: one block sums `np` partials into out[slot]
__global__ __launch_bounds__(BLK) void k_sum_partials(int np, const double *__restrict__ partial, double *__restrict__ out,
                                                       int slot) {
  __shared__ double red[BLK / 64];
  double acc = 0;
  for (int i = threadIdx.x; i < np; i += BLK) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[slot] = ((red[0] + red[1]) + red[2]) + red[3];
}

// Several sums of squares in ONE launch pair (the LM loop needs |r|^2, |gp|^2, |x_p|^2 ... one after the other: on small
// problems every one of those two-kernel launches costs more in launch latency than in work).  Vector v owns the blocks
// [v RED_BLOCKS, v RED_BLOCKS + nb_v) and sums exactly what k_sumsq would with a grid of nb_v blocks; the second kernel is
// k_sum_partials per vector, one after the other in ONE workgroup: the results are bit-identical to the separate launches.
__global__ __launch_bounds__(BLK) void k_sumsq_multi(SumsqJobs jobs, double *__restrict__ partial) {
  __shared__ double red[BLK / 64];
  const int v = blockIdx.x / RED_BLOCKS, b = blockIdx.x % RED_BLOCKS;
  const int nb = jobs.nb[v];
  if (b >= nb) return;
  const double *__restrict__ x = jobs.v[v];
  const int64_t n = jobs.n[v];
  double acc = 0;
  for (int64_t i = (int64_t)b * BLK + threadIdx.x; i < n; i += (int64_t)nb * BLK) acc += x[i] * x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[v * RED_BLOCKS + b] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(BLK) void k_sum_partials_multi(SumsqJobs jobs, const double *__restrict__ partial) {
  __shared__ double red[BLK / 64];
  for (int v = 0; v < jobs.count; v++) {  // one workgroup: the jobs one after the other (at most 1024 partials each)
    const double *pp = partial + v * RED_BLOCKS;
    double acc = 0;
    for (int i = threadIdx.x; i < jobs.nb[v]; i += BLK) acc += pp[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) jobs.out[v][jobs.slot[v]] = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
  }
  if (jobs.ha) {  // the controller's scalars to pinned host memory (thread 0 wrote the sums: ordered by the barrier)
    __threadfence();
    const int t = threadIdx.x;
    if (t < jobs.na) jobs.ha[t] = jobs.pa[t];
    if (t < jobs.nb2) jobs.hb[t] = jobs.pb[t];
    if (t == 0 && jobs.pflag) jobs.hflag[0] = jobs.pflag[0];
  }
}

__global__ __launch_bounds__(BLK) void k_axpy(int64_t n, const double *__restrict__ x, const double *__restrict__ d,
                                               double *__restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) y[i] = x[i] + d[i];
}

__global__ __launch_bounds__(BLK) void k_scale_scalar(int64_t n, double *__restrict__ v, double alpha) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) v[i] *= alpha;
}

// column scaling of the normal equations (normalize = :J / :A, src/lma_aux.jl:102-154): s_j = sqrt(diag(J'J)_j [+ lambda])
__global__ __launch_bounds__(BLK) void k_scale_vec(int64_t n, const double *__restrict__ s, double *__restrict__ v,
                                                    int divide) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n && s[i] != 0) v[i] = divide ? v[i] / s[i] : v[i] * s[i];
}

// ---- preconditioned conjugate gradients on the reduced camera system (facto = PCG) -------------------------------------
// S = Hcc + lambda I - W U^-1 W' is never formed: S v = Hcc v + lambda v - sum_a B_a' A_a h_p(a) with h = U^-1 W' v, i.e. the
// back-substitution pass (by point) followed by the right-hand-side pass (by camera) -- two sweeps over J per product.
// Preconditioner: the 9 x 9 diagonal blocks of S (block Jacobi).

// diagonal blocks: out45_c = Hcc_c - sum_{a in c} (B_a' A_a) U^-1 (B_a' A_a)'   (packed lower, row-major); one workgroup
// per camera, the fixed tree of k_cam_blocks
__global__ __launch_bounds__(BLK) void k_schur_diag(const int *__restrict__ cam_ptr, const int *__restrict__ cam_obs,
                                                     const int *__restrict__ pnt0, const double *__restrict__ J,
                                                     const double *__restrict__ Uinv, const double *__restrict__ Hcc,
                                                     double *__restrict__ out45) {
  __shared__ double red[BLK / 64][45];
  const int c = blockIdx.x;
  double acc[45];
#pragma unroll
  for (int i = 0; i < 45; i++) acc[i] = 0;
  for (int q = cam_ptr[c] + threadIdx.x; q < cam_ptr[c + 1]; q += BLK) {
    const int64_t o = cam_obs[q];
    const double *Jo = J + 24 * o;
    const double *U = Uinv + 6 * (int64_t)pnt0[o];
    const double a0[3] = {Jo[0], Jo[1], Jo[2]}, a1[3] = {Jo[12], Jo[13], Jo[14]};
    const double Um[3][3] = {{U[0], U[1], U[2]}, {U[1], U[3], U[4]}, {U[2], U[4], U[5]}};
    double T[9][3], M[9][3];
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
      for (int k = 0; k < 3; k++) T[i][k] = Jo[3 + i] * a0[k] + Jo[15 + i] * a1[k];
#pragma unroll
      for (int k = 0; k < 3; k++) M[i][k] = (T[i][0] * Um[0][k] + T[i][1] * Um[1][k]) + T[i][2] * Um[2][k];
    }
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) acc[idx++] += (M[i][0] * T[j][0] + M[i][1] * T[j][1]) + M[i][2] * T[j][2];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 45; i++) {
    double v = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 45)
    out45[45 * (int64_t)c + threadIdx.x] = Hcc[45 * (int64_t)c + threadIdx.x] -
                                            (((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// Cholesky of blk45_c + lambda I in place (one lane per camera); a non-positive pivot raises the flag
__global__ __launch_bounds__(BLK) void k_pcg_factor(int64_t ncams, double lambda, double *__restrict__ blk45, int *__restrict__ flag) {
  const int64_t c = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (c >= ncams) return;
  double L[9][9];
  int idx = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) L[i][j] = blk45[45 * c + idx++] + (i == j ? lambda : 0.0);
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    double s = L[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (!(s > 0.0)) bad = true;
    const double d = sqrt(s), inv = 1.0 / d;
    L[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < 9; i++) {
      double t = L[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t * inv;
    }
  }
  idx = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) blk45[45 * c + idx++] = L[i][j];
  if (bad) *flag = 1;
}

// The point sweep, h_p = -U_p^-1 sum_a A_a' (B_a v[cam(a)]) (the PCG product; with FULL the back-substitution of every step):
// a wave owns 64 consecutive points and walks their (contiguous) observations in batches of 64 staged rows; the
// per-observation part -- B_a v, then A_a' of it -- is done with ONE LANE PER OBSERVATION of the batch, the lanes then switch
// to their points and add up the 3-vectors of their own observations from LDS.  (The first staged version let a lane do all of
// that for its point's observations one after the other, while the lanes of the other ~50 points of the wave had no
// observation in the batch.)
template <bool FULL>
__global__ __launch_bounds__(BLK) void k_wtv(int64_t npnts, const int *__restrict__ pt_ptr, const int *__restrict__ cam0,
                                              const double *__restrict__ J, const double *__restrict__ Uinv,
                                              const double *__restrict__ v, double *__restrict__ h,
                                              const double *__restrict__ u = nullptr, const double *__restrict__ r = nullptr,
                                              double cr = 1.0, double *__restrict__ partial = nullptr) {
  // FULL: the back-substitution of the direct path, h = -(u + U^-1 W' v), and -- r != null -- the model value of the step in
  // a second sweep over the wave's rows, again one lane per observation: partial[blockIdx.x] = this block's part of
  // sum |A h + B v + cr r|^2
  __shared__ double slots[(BLK / 64) * ST_WAVE_ELEMS];
  __shared__ double wb[(BLK / 64) * 64 * 3];
  __shared__ double dpw[FULL ? (BLK / 64) * 64 * 3 : 1];
  __shared__ int own[FULL ? BLK : 1];
  __shared__ double red[BLK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * BLK + wv * 64;
  double acc = 0;
  if (p0 < npnts) {  // wave-uniform
    double *slot = slots + wv * ST_WAVE_ELEMS, *wbuf = wb + wv * 64 * 3;
    const int64_t p = p0 + lane, plast = p0 + 64 < npnts ? p0 + 64 : npnts;
    const int qb = pt_ptr[p0], qe = pt_ptr[plast];
    const int mb = p < npnts ? pt_ptr[p] : qe, me = p < npnts ? pt_ptr[p + 1] : qe;
    double t[3] = {0, 0, 0};
    d2n pv[12];
    int cn = 0;
    auto issue = [&](int q0) {  // the rows and camera indices of the next batch, requested before the current one is worked on
      const int nrows = qe - q0 < 64 ? qe - q0 : 64;
      cn = cam0[q0 + (lane < nrows ? lane : nrows - 1)];
      stage_issue(J, nullptr, q0, nrows, pv);
    };
    if (qb < qe) issue(qb);
    for (int q0 = qb; q0 < qe; q0 += 64) {
      const int nrows = qe - q0 < 64 ? qe - q0 : 64;
      const int ci = cn;
      stage_commit(pv, slot);
      if (q0 + 64 < qe) issue(q0 + 64);
      if (lane < nrows) {
        const double *Jo = slot + lane * JLD;
        const double *d = v + 9 * (int64_t)ci;
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
          s0 += Jo[3 + i] * d[i];
          s1 += Jo[15 + i] * d[i];
        }
        wbuf[3 * lane + 0] = Jo[0] * s0 + Jo[12] * s1;
        wbuf[3 * lane + 1] = Jo[1] * s0 + Jo[13] * s1;
        wbuf[3 * lane + 2] = Jo[2] * s0 + Jo[14] * s1;
      }
      wave_lds_sync();
      const int lo = mb > q0 ? mb : q0, hi = me < q0 + 64 ? me : q0 + 64;
      for (int q = lo; q < hi; q++) {
        t[0] += wbuf[3 * (q - q0) + 0];
        t[1] += wbuf[3 * (q - q0) + 1];
        t[2] += wbuf[3 * (q - q0) + 2];
      }
      wave_lds_sync();  // slot and wbuf are rewritten by the next batch
    }
    double d3[3] = {0, 0, 0};
    if (p < npnts) {
      const double *U = Uinv + 6 * p;
      if (FULL) {
        d3[0] = -(u[3 * p + 0] + ((U[0] * t[0] + U[1] * t[1]) + U[2] * t[2]));
        d3[1] = -(u[3 * p + 1] + ((U[1] * t[0] + U[3] * t[1]) + U[4] * t[2]));
        d3[2] = -(u[3 * p + 2] + ((U[2] * t[0] + U[4] * t[1]) + U[5] * t[2]));
      } else {
        d3[0] = -((U[0] * t[0] + U[1] * t[1]) + U[2] * t[2]);
        d3[1] = -((U[1] * t[0] + U[3] * t[1]) + U[4] * t[2]);
        d3[2] = -((U[2] * t[0] + U[4] * t[1]) + U[5] * t[2]);
      }
      h[3 * p + 0] = d3[0];
      h[3 * p + 1] = d3[1];
      h[3 * p + 2] = d3[2];
    }
    if (FULL && r) {
      double *dw = dpw + wv * 64 * 3;
      int *ow = own + wv * 64;
      dw[3 * lane + 0] = d3[0];
      dw[3 * lane + 1] = d3[1];
      dw[3 * lane + 2] = d3[2];
      if (qb < qe) issue(qb);
      for (int q0 = qb; q0 < qe; q0 += 64) {
        const int nrows = qe - q0 < 64 ? qe - q0 : 64;
        const int ci = cn;
        const int lo = mb > q0 ? mb : q0, hi = me < q0 + 64 ? me : q0 + 64;
        for (int q = lo; q < hi; q++) ow[q - q0] = lane;  // which lane holds the point of row q
        stage_commit(pv, slot);                          // (its fence also publishes ow and, the first time, dw)
        if (q0 + 64 < qe) issue(q0 + 64);
        if (lane < nrows) {
          const double *Jo = slot + lane * JLD;
          const double *d = v + 9 * (int64_t)ci;
          const double *dq = dw + 3 * ow[lane];
          const int64_t q = q0 + lane;
          double s0 = cr * r[2 * q], s1 = cr * r[2 * q + 1];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            s0 += Jo[i] * dq[i];
            s1 += Jo[12 + i] * dq[i];
          }
#pragma unroll
          for (int i = 0; i < 9; i++) {
            s0 += Jo[3 + i] * d[i];
            s1 += Jo[15 + i] * d[i];
          }
          acc += s0 * s0 + s1 * s1;
        }
        wave_lds_sync();
      }
    }
  }
  if (FULL && r) {
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
  }
}

// Hpp (6) and gp (3) per point for point-sorted observations, one lane per OBSERVATION for the nine products and one lane
// per point for their sum (k_wtv's scheme; only the point part of a row is needed: plain 8-byte loads, no staging).
__global__ __launch_bounds__(BLK) void k_point_blocks_lpo(int64_t npnts, const int *__restrict__ pt_ptr, const double *__restrict__ J,
                                                           const double *__restrict__ r, double *__restrict__ Hpp,
                                                           double *__restrict__ gp) {
  __shared__ double vb[(BLK / 64) * 64 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * BLK + wv * 64;
  if (p0 >= npnts) return;  // wave-uniform; no workgroup barrier below
  double *vw = vb + wv * 64 * 9;
  const int64_t p = p0 + lane, plast = p0 + 64 < npnts ? p0 + 64 : npnts;
  const int qb = pt_ptr[p0], qe = pt_ptr[plast];
  const int mb = p < npnts ? pt_ptr[p] : qe, me = p < npnts ? pt_ptr[p + 1] : qe;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q0 = qb; q0 < qe; q0 += 64) {
    const int nrows = qe - q0 < 64 ? qe - q0 : 64;
    if (lane < nrows) {
      const int64_t o = q0 + lane;
      const double *Jo = J + 24 * o;
      const double a0[3] = {Jo[0], Jo[1], Jo[2]}, a1[3] = {Jo[12], Jo[13], Jo[14]};
      double *w = vw + 9 * lane;
      w[0] = a0[0] * a0[0] + a1[0] * a1[0];
      w[1] = a0[0] * a0[1] + a1[0] * a1[1];
      w[2] = a0[0] * a0[2] + a1[0] * a1[2];
      w[3] = a0[1] * a0[1] + a1[1] * a1[1];
      w[4] = a0[1] * a0[2] + a1[1] * a1[2];
      w[5] = a0[2] * a0[2] + a1[2] * a1[2];
      if (gp) {
        const double r0 = r[2 * o], r1 = r[2 * o + 1];
        w[6] = a0[0] * r0 + a1[0] * r1;
        w[7] = a0[1] * r0 + a1[1] * r1;
        w[8] = a0[2] * r0 + a1[2] * r1;
      }
    }
    wave_lds_sync();
    const int lo = mb > q0 ? mb : q0, hi = me < q0 + 64 ? me : q0 + 64;
    for (int q = lo; q < hi; q++) {
      const double *w = vw + 9 * (q - q0);
#pragma unroll
      for (int i = 0; i < 6; i++) acc[i] += w[i];
      if (gp) {
#pragma unroll
        for (int i = 6; i < 9; i++) acc[i] += w[i];
      }
    }
    wave_lds_sync();
  }
  if (p < npnts) {
    if (Hpp)
      for (int i = 0; i < 6; i++) Hpp[6 * p + i] = acc[i];
    if (gp)
      for (int i = 0; i < 3; i++) gp[3 * p + i] = acc[6 + i];
  }
}

__global__ __launch_bounds__(BLK) void k_gather_int(int64_t n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// y += a x
__global__ __launch_bounds__(BLK) void k_axpy_s(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

// The scalars of the CG iteration live on the device (cg[0] = p.q, cg[1] = r.r, cg[2] = r.z, cg[3] = alpha, cg[4] = beta):
// the vector kernels read alpha / beta from there and the host looks at them once per iteration, for the stopping test only.
// Reductions by ONE workgroup of 1024 lanes with a fixed tree (n = 9 ncams <= 1.3e5): deterministic and one launch.
__device__ inline double block1024_sum(double v, double *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x < 16) t = red[threadIdx.x];
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 4, 64);
  t += __shfl_xor(t, 8, 64);
  __syncthreads();
  return t;  // valid in lane 0 of wave 0
}
// alpha = r.z / p.q (0 when p.q is not positive: the host stops there)
__global__ __launch_bounds__(1024) void k_cg_alpha(int64_t n, const double *__restrict__ pd, const double *__restrict__ q,
                                                    double *__restrict__ cg) {
  __shared__ double red[16];
  double acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += pd[i] * q[i];
  const double pq = block1024_sum(acc, red);
  if (threadIdx.x == 0) {
    cg[0] = pq;
    cg[3] = pq > 0 ? cg[2] / pq : 0.0;
  }
}
// x += alpha p, r -= alpha q, z = (L L')^-1 r per camera; per-block partial sums of r.r and r.z (one lane per camera).
// first != 0: the start of the solve (x = 0, r = rhs are given): only z and the partial sums.
__global__ __launch_bounds__(BLK) void k_cg_step(int64_t ncams, const double *__restrict__ cg, const double *__restrict__ L45,
                                                  const double *__restrict__ pd, const double *__restrict__ q, double *__restrict__ x,
                                                  double *__restrict__ r, double *__restrict__ z, double *__restrict__ partial,
                                                  int first) {
  __shared__ double red[2][BLK / 64];
  const int64_t c = (int64_t)blockIdx.x * BLK + threadIdx.x;
  double rr = 0, rz = 0;
  if (c < ncams) {
    const double alpha = first ? 0.0 : cg[3];
    double L[9][9], y[9], rc[9];
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) L[i][j] = L45[45 * c + idx++];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      if (first) {
        rc[i] = r[9 * c + i];
      } else {
        x[9 * c + i] += alpha * pd[9 * c + i];
        rc[i] = r[9 * c + i] - alpha * q[9 * c + i];
        r[9 * c + i] = rc[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      double t = rc[i];
#pragma unroll
      for (int k = 0; k < i; k++) t -= L[i][k] * y[k];
      y[i] = t / L[i][i];
    }
#pragma unroll
    for (int i = 8; i >= 0; i--) {
      double t = y[i];
#pragma unroll
      for (int k = i + 1; k < 9; k++) t -= L[k][i] * y[k];
      y[i] = t / L[i][i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      z[9 * c + i] = y[i];
      rr += rc[i] * rc[i];
      rz += rc[i] * y[i];
    }
  }
  rr = wave_sum(rr);
  rz = wave_sum(rz);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = rr;
    red[1][threadIdx.x >> 6] = rz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    partial[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}
// r.r, r.z from the partials; beta = r.z / (previous r.z); p = z + beta p (first: p = z).  One workgroup.
__global__ __launch_bounds__(1024) void k_cg_beta_dir(int64_t n, int nb, const double *__restrict__ partial, double *__restrict__ cg,
                                                       const double *__restrict__ z, double *__restrict__ pd, int first) {
  __shared__ double red[16];
  __shared__ double sbeta;
  double a0 = 0, a1 = 0;
  for (int i = threadIdx.x; i < nb; i += 1024) {
    a0 += partial[2 * i];
    a1 += partial[2 * i + 1];
  }
  const double rr = block1024_sum(a0, red);
  const double rz = block1024_sum(a1, red);
  if (threadIdx.x == 0) {
    const double beta = first ? 0.0 : rz / cg[2];
    cg[1] = rr;
    cg[2] = rz;
    cg[4] = beta;
    sbeta = beta;
  }
  __syncthreads();
  const double beta = sbeta;
  for (int64_t i = threadIdx.x; i < n; i += 1024) pd[i] = first ? z[i] : z[i] + beta * pd[i];
}

}  // namespace


static inline unsigned grid_for(int64_t n, int blk) { return (unsigned)((n + blk - 1) / blk); }

// BA_STAGED=0: the scalar-load kernels of round 1 everywhere (A/B measurements)
static bool staged_on() {
  static const bool off = [] { const char *e = getenv("BA_STAGED"); return e && e[0] == '0'; }();
  return !off;
}

int launch_point_blocks(ba_problem *p, const double *d_J, const double *d_r, double *d_Hpp, double *d_gp,
                        hipStream_t st) {
  if (p->npnts == 0) return BA_OK;
  ProfScope ps(p, PC_POINT_BLOCKS, st);
  if (p->point_sorted)
    hipLaunchKernelGGL(k_point_blocks_lpo, dim3(grid_for(p->npnts, BLK)), dim3(BLK), 0, st, p->npnts, p->pt_ptr, d_J, d_r, d_Hpp, d_gp);
  else
    hipLaunchKernelGGL(k_point_blocks, dim3(grid_for(p->npnts, BLK)), dim3(BLK), 0, st, p->npnts, p->pt_ptr, p->pt_obs,
                       d_J, d_r, d_Hpp, d_gp);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_cam_blocks(ba_problem *p, const double *d_J, const double *d_r, double *d_Hcc, double *d_gc,
                      hipStream_t st) {
  if (p->ncams == 0) return BA_OK;
  ProfScope ps(p, PC_CAM_BLOCKS, st);
  // the 54-accumulator Hcc pass is faster with plain per-lane loads (0.32 against 0.50 ms on Venice); the light passes
  // (J'r here, the right-hand side in launch_schur_rhs) are latency-bound and gain from the staged rows (0.68 -> 0.47 ms)
  if (d_Hcc)
    hipLaunchKernelGGL(k_cam_blocks<0>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_r, (const double *)nullptr, d_Hcc, d_gc);
  else if (staged_on())
    hipLaunchKernelGGL(k_cam_blocks_st<1>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_r, (const double *)nullptr, (double *)nullptr, d_gc);
  else
    hipLaunchKernelGGL(k_cam_blocks<1>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_r, (const double *)nullptr, (double *)nullptr, d_gc);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_schur_prep(ba_problem *p, double lambda, const double *d_Hpp, const double *d_gp, double *d_Uinv,
                      double *d_u, hipStream_t st, const double *d_lambda, const double *d_damp, double *d_lambda_copy,
                      double *d_zero, int64_t nzero) {
  if (p->npnts == 0) {
    if (d_lambda_copy) BA_HIP_CHECK(hipMemcpyAsync(d_lambda_copy, d_lambda, sizeof(double), hipMemcpyDefault, st));
    if (nzero > 0) BA_HIP_CHECK(hipMemsetAsync(d_zero, 0, (size_t)nzero * sizeof(double), st));
    return BA_OK;
  }
  ProfScope ps(p, PC_SCHUR_PREP, st);
  hipLaunchKernelGGL(k_schur_prep, dim3(grid_for(p->npnts, BLK)), dim3(BLK), 0, st, p->npnts, lambda, d_lambda, d_Hpp,
                     d_gp, d_Uinv, d_u, d_damp, d_lambda_copy, d_zero, nzero);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_schur_blocks(ba_problem *p, const SchurTasks *T, const double *d_J, const double *d_Uinv, double *d_Y,
                        const double *d_Hcc, double lambda, double *d_S, const int64_t *d_col_off, int64_t n, int64_t npad,
                        hipStream_t st, const double *d_lambda, const double *d_damp, int64_t s_tiles) {
  ProfScope ps(p, PC_SCHUR_S, st);
  // s_tiles: the tiles S holds (the whole lower triangle, or only the pattern's with compressed block-sparse storage)
  BA_HIP_CHECK(hipMemsetAsync(d_S, 0, (size_t)(s_tiles > 0 ? s_tiles * NB * NB : dense_ldl_tiles_doubles(n)) * sizeof(double), st));
  if (p->nobs > 0)
    hipLaunchKernelGGL(k_obs_y, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->pnt0, d_J, d_Uinv, d_Y);
  if (T->nkeys > 0) {
    int64_t nb = (T->nkeys + BLK / 64 - 1) / (BLK / 64);
    if (nb > (int64_t)1 << 22) nb = (int64_t)1 << 22;  // 2^22 blocks x 256 lanes = 2^30 work-items
    const double *damp_c = d_damp ? d_damp + 3 * p->npnts : (const double *)nullptr;
    hipLaunchKernelGGL(k_schur_blocks, dim3((unsigned)nb), dim3(BLK), 0, st, T->nkeys, T->key_ptr,
                       T->key_ca, T->key_cb, T->task_a, T->task_b, d_J, d_Y, d_Hcc, lambda, d_lambda, d_S, d_col_off, damp_c,
                       T->nsplit > 0 ? 2 * T->chunk : 0, (const int *)nullptr, T->cam_of);
    if (T->nsplit > 0) {  // long keys: chunk partials, then their fixed-order sums
      int64_t nbc = (T->nchunks + BLK / 64 - 1) / (BLK / 64), nbs = (T->nsplit + BLK / 64 - 1) / (BLK / 64);
      if (nbc > (int64_t)1 << 22) nbc = (int64_t)1 << 22;
      if (nbs > (int64_t)1 << 22) nbs = (int64_t)1 << 22;
      hipLaunchKernelGGL(k_schur_chunks, dim3((unsigned)nbc), dim3(BLK), 0, st, T->nchunks, T->chunk_t0, T->chunk_t1, T->task_a,
                         T->task_b, d_J, d_Y, T->partial);
      hipLaunchKernelGGL(k_schur_combine, dim3((unsigned)nbs), dim3(BLK), 0, st, T->nsplit, T->skey, T->skey_c0, T->key_ca,
                         T->key_cb, T->partial, d_Hcc, lambda, d_lambda, d_S, d_col_off, damp_c, (const int *)nullptr, T->cam_of);
    }
  }
  if (npad > n) hipLaunchKernelGGL(k_pad_diag, dim3(grid_for(npad - n, BLK)), dim3(BLK), 0, st, n, npad, d_S, d_col_off);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// Chunked assembly (per-rank ownership of S on several ranks): launch_schur_pre once per step -- the per-observation Y
// blocks and the partial sums of the split keys -- then launch_schur_chunk per chunk of tile columns into `dest` (this
// rank's own tiles, or the staging buffer whose content is reduced onto the chunk's owner), through the chunk's offset table.
int launch_schur_pre(ba_problem *p, const SchurTasks *T, const double *d_J, const double *d_Uinv, double *d_Y, hipStream_t st) {
  ProfScope ps(p, PC_SCHUR_S, st);
  if (p->nobs > 0)
    hipLaunchKernelGGL(k_obs_y, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->pnt0, d_J, d_Uinv, d_Y);
  if (T->nsplit > 0) {
    int64_t nbc = (T->nchunks + BLK / 64 - 1) / (BLK / 64);
    if (nbc > (int64_t)1 << 22) nbc = (int64_t)1 << 22;
    hipLaunchKernelGGL(k_schur_chunks, dim3((unsigned)nbc), dim3(BLK), 0, st, T->nchunks, T->chunk_t0, T->chunk_t1, T->task_a,
                       T->task_b, d_J, d_Y, T->partial);
  }
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_schur_chunk(ba_problem *p, const SchurTasks *T, const SchurChunk *c, const double *d_J, const double *d_Y,
                       const double *d_Hcc, double lambda, double *dest, int64_t n, int64_t npad, hipStream_t st,
                       const double *d_lambda, const double *d_damp) {
  ProfScope ps(p, PC_SCHUR_S, st);
  BA_HIP_CHECK(hipMemsetAsync(dest, 0, (size_t)c->ntiles * NB * NB * sizeof(double), st));
  const double *damp_c = d_damp ? d_damp + 3 * p->npnts : (const double *)nullptr;
  if (c->nkeys > 0) {
    int64_t nb = (c->nkeys + BLK / 64 - 1) / (BLK / 64);
    if (nb > (int64_t)1 << 22) nb = (int64_t)1 << 22;
    hipLaunchKernelGGL(k_schur_blocks, dim3((unsigned)nb), dim3(BLK), 0, st, c->nkeys, T->key_ptr, T->key_ca, T->key_cb, T->task_a,
                       T->task_b, d_J, d_Y, d_Hcc, lambda, d_lambda, dest, c->cco, damp_c, T->nsplit > 0 ? 2 * T->chunk : 0, c->keys, T->cam_of);
  }
  if (c->nskeys > 0) {
    int64_t nbs = (c->nskeys + BLK / 64 - 1) / (BLK / 64);
    if (nbs > (int64_t)1 << 22) nbs = (int64_t)1 << 22;
    hipLaunchKernelGGL(k_schur_combine, dim3((unsigned)nbs), dim3(BLK), 0, st, c->nskeys, T->skey, T->skey_c0, T->key_ca, T->key_cb,
                       T->partial, d_Hcc, lambda, d_lambda, dest, c->cco, damp_c, c->skeys, T->cam_of);
  }
  if (npad > n) hipLaunchKernelGGL(k_pad_diag, dim3(grid_for(npad - n, BLK)), dim3(BLK), 0, st, n, npad, dest, c->cco);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// S_ij /= d_i d_j on the tile columns a rank owns (own_cols / own_pref as k_ldl_update's OWN mode; block = one tile)
__global__ __launch_bounds__(BLK) void k_scale_S_own(int64_t n, const double *__restrict__ dsc, double *__restrict__ S,
                                                      const int64_t *__restrict__ co, const int *__restrict__ own_cols,
                                                      const int64_t *__restrict__ own_pref, int ncols) {
  const int64_t t = blockIdx.x;
  int lo = 0, hi = ncols;  // largest m with own_pref[m] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (own_pref[mid] <= t) lo = mid;
    else hi = mid;
  }
  const int64_t tj = own_cols[lo], ti = tj + (t - own_pref[lo]);
  double *T = S + tix(co, ti, tj) * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += BLK) {
    const int64_t r = ti * NB + (e >> 7), c = tj * NB + (e & (NB - 1));
    const double dr = r < n ? dsc[r] : 1.0, dc = c < n ? dsc[c] : 1.0;
    T[e] /= dr * dc;
  }
}
int launch_scale_S_own(ba_problem *p, int64_t n, const double *d_dsc, double *d_S, const int64_t *d_col_off, const int *d_own_cols,
                       const int64_t *d_own_pref, int ncols, int64_t ntiles, hipStream_t st) {
  if (ntiles <= 0 || ncols <= 0) return BA_OK;
  hipLaunchKernelGGL(k_scale_S_own, dim3((unsigned)ntiles), dim3(BLK), 0, st, n, d_dsc, d_S, d_col_off, d_own_cols, d_own_pref, ncols);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// the same over a list of stored tiles (block-sparse S with per-rank ownership): tile t of S is (tiles[t].x, tiles[t].y)
__global__ __launch_bounds__(BLK) void k_scale_S_list(int64_t n, const double *__restrict__ dsc, double *__restrict__ S,
                                                       const int2 *__restrict__ tiles) {
  const int64_t t = blockIdx.x, ti = tiles[t].x, tj = tiles[t].y;
  double *T = S + t * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += BLK) {
    const int64_t r = ti * NB + (e >> 7), c = tj * NB + (e & (NB - 1));
    const double dr = r < n ? dsc[r] : 1.0, dc = c < n ? dsc[c] : 1.0;
    T[e] /= dr * dc;
  }
}
int launch_scale_S_list(ba_problem *p, int64_t n, const double *d_dsc, double *d_S, const int2 *d_tiles, int64_t ntiles, hipStream_t st) {
  if (ntiles <= 0) return BA_OK;
  hipLaunchKernelGGL(k_scale_S_list, dim3((unsigned)ntiles), dim3(BLK), 0, st, n, d_dsc, d_S, d_tiles);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_hcc_diag(ba_problem *p, const double *d_Hcc, double *d_hdiag, hipStream_t st) {
  if (p->ncams == 0) return BA_OK;
  hipLaunchKernelGGL(k_hcc_diag, dim3(grid_for(9 * p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_Hcc, d_hdiag);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_cam_scale(ba_problem *p, const double *d_hdiag, double add, double *d_dsc, hipStream_t st,
                     const double *d_lambda, const int *d_pos) {
  if (p->ncams == 0) return BA_OK;
  hipLaunchKernelGGL(k_cam_scale, dim3(grid_for(9 * p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_hdiag, add, d_lambda,
                     d_dsc, d_pos);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// dst[9 c + j] = src[9 pos[c] + j]: the camera part of the step back from the block rows of S to camera order
__global__ __launch_bounds__(BLK) void k_gather_cams(int64_t ncams, const int *__restrict__ pos, const double *__restrict__ src,
                                                      double *__restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < 9 * ncams) dst[i] = src[9 * (int64_t)pos[i / 9] + i % 9];
}
int launch_gather_cams(ba_problem *p, const int *d_pos, const double *d_src, double *d_dst, hipStream_t st) {
  if (p->ncams == 0) return BA_OK;
  hipLaunchKernelGGL(k_gather_cams, dim3(grid_for(9 * p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_pos, d_src, d_dst);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_scale_S(ba_problem *p, int64_t n, int64_t nt, const double *d_dsc, double *d_S, const int64_t *d_col_off,
                   hipStream_t st) {
  hipLaunchKernelGGL(k_scale_S, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(BLK), 0, st, n, nt, d_dsc, d_S, d_col_off);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_schur_rhs(ba_problem *p, const double *d_J, const double *d_r, const double *d_u, double *d_rhs,
                     hipStream_t st, const int *d_cam_pnt, const int *d_pos) {
  if (p->ncams == 0) return BA_OK;
  ProfScope ps(p, PC_SCHUR_RHS, st);
  if (staged_on())
    hipLaunchKernelGGL(k_cam_blocks_st<2>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_r, d_u, (double *)nullptr, d_rhs, 0.0, d_cam_pnt, d_pos);
  else
    hipLaunchKernelGGL(k_cam_blocks<2>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_r, d_u, (double *)nullptr, d_rhs, 0.0, d_pos);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// d_r_model != null (and the staged kernel applies): |J delta + cr r|^2 of the step is formed in the same pass ->
// d_scal[slot], *model_done = true.  d_partial then needs one entry per block of 256 points.
int launch_backsub(ba_problem *p, const double *d_J, const double *d_Uinv, const double *d_u, const double *d_dc,
                   double *d_dp, hipStream_t st, const double *d_r_model, double cr, double *d_partial, double *d_scal,
                   int slot, bool *model_done) {
  if (model_done) *model_done = false;
  if (p->npnts == 0) return BA_OK;
  ProfScope ps(p, PC_BACKSUB, st);
  const unsigned nb = grid_for(p->npnts, BLK);
  if (p->point_sorted && staged_on()) {
    const bool with_model = d_r_model && d_partial && d_scal;
    // one lane per observation in both sweeps (k_wtv<true>; the first staged version walked a point's observations with one
    // lane: Venice 0.75 -> 0.33 ms, Dubrovnik 0.28 -> 0.10 ms per call)
    hipLaunchKernelGGL(k_wtv<true>, dim3(nb), dim3(BLK), 0, st, p->npnts, p->pt_ptr, p->cam0, d_J, d_Uinv, d_dc, d_dp, d_u,
                       with_model ? d_r_model : (const double *)nullptr, cr, d_partial);
    if (with_model) {
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLK), 0, st, (int)nb, d_partial, d_scal, slot);
      if (model_done) *model_done = true;
    }
  } else {
    hipLaunchKernelGGL(k_backsub, dim3(nb), dim3(BLK), 0, st, p->npnts, p->pt_ptr, p->pt_obs, p->cam0, d_J, d_Uinv, d_u, d_dc,
                       d_dp);
  }
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// |J delta + cr r|^2 -> scal[slot]
int launch_model_sq(ba_problem *p, const double *d_J, const double *d_r, const double *d_delta, double *d_partial,
                    double *d_scal, int slot, hipStream_t st, double cr) {
  ProfScope ps(p, PC_TRIAL, st);
  int nb = (int)((p->nobs + BLK - 1) / BLK);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_model_sq, dim3(nb), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0, p->pnt0, d_J, d_r, d_delta,
                     cr, d_partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLK), 0, st, nb, d_partial, d_scal, slot);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// |v|^2 -> scal[slot]
int launch_sumsq(ba_problem *p, int64_t n, const double *d_v, double *d_partial, double *d_scal, int slot,
                 hipStream_t st) {
  ProfScope ps(p, PC_REDUCE, st);
  int nb = (int)((n + BLK - 1) / BLK);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(BLK), 0, st, n, d_v, d_partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLK), 0, st, nb, d_partial, d_scal, slot);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_sumsq_multi(ba_problem *p, SumsqJobs *jobs, double *d_partial_multi, hipStream_t st) {
  if (jobs->count <= 0) return BA_OK;
  ProfScope ps(p, PC_REDUCE, st);
  for (int v = 0; v < jobs->count; v++) {
    int nb = (int)((jobs->n[v] + BLK - 1) / BLK);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    jobs->nb[v] = nb;
  }
  hipLaunchKernelGGL(k_sumsq_multi, dim3(jobs->count * RED_BLOCKS), dim3(BLK), 0, st, *jobs, d_partial_multi);
  hipLaunchKernelGGL(k_sum_partials_multi, dim3(1), dim3(BLK), 0, st, *jobs, (const double *)d_partial_multi);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// The LM controller's scalars to pinned host memory by a kernel that addresses the host buffers directly (hipHostMalloc
// memory is mapped): in a recorded launch sequence each of the former three copy nodes cost a dispatch of ~5 us of its own.
// (Recorded sequences publish from their last reduction kernel instead, k_sum_partials_multi; the damping travels the other
// way inside k_schur_prep.)
__global__ void k_publish(const double *__restrict__ a, int na, double *__restrict__ ha, const double *__restrict__ b, int nb,
                          double *__restrict__ hb, const int *__restrict__ flag, int *__restrict__ hflag) {
  const int t = threadIdx.x;
  if (t < na) ha[t] = a[t];
  if (t < nb) hb[t] = b[t];
  if (t == 0 && flag) hflag[0] = flag[0];
}

int launch_publish(const double *d_a, int na, double *h_a, const double *d_b, int nb, double *h_b, const int *d_flag, int *h_flag,
                   hipStream_t st) {
  if (na > 64 || nb > 64) {
    ba_set_error("launch_publish: more than 64 scalars");
    return BA_ERR_ARG;
  }
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d_a, na, h_a, d_b, nb, h_b, d_flag, h_flag);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_axpy(ba_problem *p, int64_t n, const double *d_x, const double *d_d, double *d_y, hipStream_t st) {
  if (n == 0) return BA_OK;
  hipLaunchKernelGGL(k_axpy, dim3(grid_for(n, BLK)), dim3(BLK), 0, st, n, d_x, d_d, d_y);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_scale_scalar(ba_problem *p, int64_t n, double *d_v, double alpha, hipStream_t st) {
  if (n == 0) return BA_OK;
  hipLaunchKernelGGL(k_scale_scalar, dim3(grid_for(n, BLK)), dim3(BLK), 0, st, n, d_v, alpha);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_scale_vec(ba_problem *p, int64_t n, const double *d_s, double *d_v, int divide, hipStream_t st) {
  if (n == 0) return BA_OK;
  hipLaunchKernelGGL(k_scale_vec, dim3(grid_for(n, BLK)), dim3(BLK), 0, st, n, d_s, d_v, divide);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// ---- facto_type = Float16 helpers (see k_f16_cols) ---------------------------------------------------------------------
int launch_col_sq(ba_problem *p, const double *d_Hpp, const double *d_hdiag, double *d_jn2, hipStream_t st) {
  const int64_t nvar = 3 * p->npnts + 9 * p->ncams;
  if (nvar == 0) return BA_OK;
  hipLaunchKernelGGL(k_col_sq, dim3(grid_for(nvar, BLK)), dim3(BLK), 0, st, p->npnts, p->ncams, d_Hpp, d_hdiag, d_jn2);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

int launch_f16_scale(ba_problem *p, double lambda, double mu, const double *d_jn2, const double *d_J, const double *d_r,
                     double *d_dcol, double *d_damp, double *d_Jq, double *d_rq, hipStream_t st) {
  const int64_t nvar = 3 * p->npnts + 9 * p->ncams;
  if (nvar > 0) hipLaunchKernelGGL(k_f16_cols, dim3(grid_for(nvar, BLK)), dim3(BLK), 0, st, nvar, d_jn2, lambda, mu, d_dcol, d_damp);
  if (p->nobs > 0)
    hipLaunchKernelGGL(k_f16_quantize, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->npnts, p->cam0, p->pnt0, d_J,
                       d_r, d_dcol, mu, d_Jq, d_rq);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}

// ---- PCG launchers ------------------------------------------------------------------------------------------------------
// q_c = Hcc_c v_c + lam v_c + sum_a B_a' A_a h_p(a)   (the camera sweep of the PCG product)
int launch_wuw(ba_problem *p, const double *d_J, const double *d_h, const double *d_Hcc, const double *d_v, double lam, double *d_q,
               hipStream_t st, const int *d_cam_pnt) {
  if (p->ncams == 0) return BA_OK;
  ProfScope ps(p, PC_SCHUR_RHS, st);
  if (staged_on())
    hipLaunchKernelGGL(k_cam_blocks_st<3>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_v, d_h, const_cast<double *>(d_Hcc), d_q, lam, d_cam_pnt);
  else
    hipLaunchKernelGGL(k_cam_blocks<3>, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J,
                       d_v, d_h, const_cast<double *>(d_Hcc), d_q, lam);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
int launch_schur_diag(ba_problem *p, const double *d_J, const double *d_Uinv, const double *d_Hcc, double *d_blk45, hipStream_t st) {
  if (p->ncams == 0) return BA_OK;
  ProfScope ps(p, PC_SCHUR_S, st);
  hipLaunchKernelGGL(k_schur_diag, dim3((unsigned)p->ncams), dim3(BLK), 0, st, p->cam_ptr, p->cam_obs, p->pnt0, d_J, d_Uinv, d_Hcc,
                     d_blk45);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
int launch_pcg_factor(ba_problem *p, double lambda, double *d_blk45, int *d_flag, hipStream_t st) {
  hipLaunchKernelGGL(k_pcg_factor, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, lambda, d_blk45, d_flag);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
int launch_axpy_s(ba_problem *p, int64_t n, double a, const double *d_x, double *d_y, hipStream_t st) {
  if (n == 0) return BA_OK;
  hipLaunchKernelGGL(k_axpy_s, dim3(grid_for(n, BLK)), dim3(BLK), 0, st, n, a, d_x, d_y);
  return BA_OK;
}
int launch_cg_alpha(ba_problem *p, int64_t n, const double *d_p, const double *d_q, double *d_cg, hipStream_t st) {
  ProfScope ps(p, PC_REDUCE, st);
  hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(1024), 0, st, n, d_p, d_q, d_cg);
  return BA_OK;
}
// d_partial: 2 entries per block of 256 cameras
int launch_cg_step(ba_problem *p, const double *d_cg, const double *d_L45, const double *d_p, const double *d_q, double *d_x,
                   double *d_r, double *d_z, double *d_partial, int first, hipStream_t st) {
  if (p->ncams == 0) return BA_OK;
  hipLaunchKernelGGL(k_cg_step, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_cg, d_L45, d_p, d_q, d_x, d_r, d_z,
                     d_partial, first);
  return BA_OK;
}
int launch_cg_beta_dir(ba_problem *p, int64_t n, const double *d_partial, double *d_cg, const double *d_z, double *d_p, int first,
                       hipStream_t st) {
  ProfScope ps(p, PC_REDUCE, st);
  hipLaunchKernelGGL(k_cg_beta_dir, dim3(1), dim3(1024), 0, st, n, (int)grid_for(p->ncams, BLK), d_partial, d_cg, d_z, d_p, first);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
// h = -U^-1 W' v (the point sweep of the PCG product); observations grouped by point (BAL order)
int launch_wtv(ba_problem *p, const double *d_J, const double *d_Uinv, const double *d_v, double *d_h, hipStream_t st) {
  if (p->npnts == 0) return BA_OK;
  ProfScope ps(p, PC_BACKSUB, st);
  hipLaunchKernelGGL(k_wtv<false>, dim3(grid_for(p->npnts, BLK)), dim3(BLK), 0, st, p->npnts, p->pt_ptr, p->cam0, d_J, d_Uinv, d_v, d_h);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
// cam_pnt[q] = pnt0[cam_obs[q]]: the point of every observation in camera order (built once per problem)
int launch_cam_pnt(ba_problem *p, int *d_cam_pnt, hipStream_t st) {
  if (p->nobs == 0) return BA_OK;
  hipLaunchKernelGGL(k_gather_int, dim3(grid_for(p->nobs, BLK)), dim3(BLK), 0, st, p->nobs, p->cam_obs, p->pnt0, d_cam_pnt);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
