// Bench-only entry points of the dense LDL' kernels (tools/bench_update.py, bench_diag.py, bench_mfma_probe.py): built into
// tools/libba_bench.so, NOT into libba_hip.so.  This translation unit includes the product source so that the probes
// time the very kernels that ship.
#include <time.h>
#include "../ba_dense_ldl.hip"

// ---- probe variants of the bulk trailing update (micro-benchmark only: tools/bench_update.py) -----------------------------
// DBG: bit 0 = store instead of read-modify-write, bit 1 = every workgroup reads the same operand tiles (L2-resident
// operands), bit 3 = workgroup-shared LDS staging with barriers (the first version: 52.6 TFLOP/s against 59.8 for the
// wave-private staging that ships), bit 4 = LDS-DMA staging, bit 5 = K = 512 (four panels per pass), bit 6 = accumulators
// started from -C, bit 7 = direct operand loads without LDS (bits 8-12: its sub-probes), bit 13 = super-block enumeration of the
// triangle (adopted in the product kernel).  DBG = 0 is the product kernel k_ldl_update itself.
namespace {
// LDS-DMA variant of the wave-private product (Float64): the operand chunks go global -> LDS directly
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write, no wait between a load and its LDS write), chunks of 8, two LDS
// buffers per wave (16 KB per wave, 64 KB per workgroup); the next chunk's 8 pieces are in flight during the MFMAs and the
// wave waits with vmcnt only.  A piece lands at base + lane * 16 bytes, so rows are unpadded (64 B): the 16-byte columns are
// XOR-swizzled with the row ((row >> 2) & 3), which keeps the MFMA operand reads at two dwords per bank.  Same products in
// the same order as tile_gemm_abt_priv (bit-identical accumulators; tools/bench_mfma_probe.py modes 3 / 5 compare their
// checksums).  Probe, nonzero operands, steady state: 68.6-68.7 TFLOP/s against 66.1-66.2 for the register-staged loop.
// Direct-operand variant of the wave-private product (probe; bits 7-12 of DBG): NO LDS.  The A / B operand of the 16x16x4 matrix instruction is "lane l holds element
// [row l & 15][k = l >> 4]" of a K-contiguous row -- exactly what a lane reads from a row-major tile, so the wave-private
// staging above passes every operand byte through LDS (a 16-byte write and an 8-byte read per element pair) without
// sharing or transposing anything.  Here a lane loads 16 bytes of ITS row straight into registers: VL = 2 doubles (4 floats)
// that feed VL consecutive matrix instructions.  A group = 4 VL consecutive k (64 bytes of every row: the 4 lanes of a row
// cover it); lane (fr, fk) holds k = g 4 VL + VL fk + s for step s of group g.  Within a step the four fk lanes thus carry
// k = 4VL g + s, + VL, + 2VL, + 3VL instead of four consecutive k: a permutation of the K sum (same products, another
// order: adopting it would need the same permutation -- kperm -- in the row-split update, whose bits must equal this kernel's).
// Three register slots of 8 loads each: group g is multiplied while g + 1 and g + 2 are in flight.
// Measured (profiles/r04_f_update_probes.txt): the same speed as the LDS-staged product kernel (58.7 against 60.0 TFLOP/s on
// zero operands, 57.7 against 56.6 on full-entropy ones) -- the staging is not what the kernel waits for.  PROBE bits:
// 1 half the loads (+11 %), 2 L1-resident operands (+11 %), 4 / 8 / 16 workgroup barriers that keep the waves in step (0 %).
template <typename T>
__device__ __forceinline__ int kperm(int s4, int fk) {  // the k (within a panel) of matrix-instruction step s4, lane group fk
  constexpr int VL = 16 / sizeof(T);
  return (s4 / VL) * 4 * VL + VL * fk + (s4 % VL);
}
template <typename T, int NP, int PROBE = 0>
__device__ inline void tile_gemm_abt_direct(const T *__restrict__ A0, const T *__restrict__ B0,
                                            const T *__restrict__ A1, const T *__restrict__ B1,
                                            typename RT<T>::v4 acc[4][4], const T *__restrict__ A2 = nullptr,
                                            const T *__restrict__ B2 = nullptr, const T *__restrict__ A3 = nullptr,
                                            const T *__restrict__ B3 = nullptr) {
  typedef typename PV<T>::vu vu;
  constexpr int VL = PV<T>::VL, GK = 4 * VL, GPP = NB / GK, NG = NP * GPP, R = 3;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  const int fr = lane & 15, fk = lane >> 4;
  uint32_t voff[4];  // the lane's byte offset inside the 64-row slice, per 16-row block (uniform bases: one VGPR each)
#pragma unroll
  for (int m = 0; m < 4; m++) {
    voff[m] = (uint32_t)(((16 * m + fr) * NB + VL * fk) * sizeof(T));
    asm volatile("" : "+v"(voff[m]));
  }
  vu ra[R][4], rb[R][4];
  // the tile addresses are the same for every lane of the workgroup, but come out of per-lane arithmetic (the triangular
  // index): told so, the compiler keeps them in scalar registers and every load is base (scalar) + lane offset + immediate
  auto uni = [](const T *q) {
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    return reinterpret_cast<const T *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
  };
  typedef __attribute__((address_space(1))) char gchar;  // (global address space: the rebuilt pointers would be generic)
  typedef __attribute__((address_space(1))) vu gvu;
  const T *Ap[4] = {uni(A0), uni(A1), NP > 2 ? uni(A2) : nullptr, NP > 2 ? uni(A3) : nullptr};
  const T *Bp[4] = {uni(B0), uni(B1), NP > 2 ? uni(B2) : nullptr, NP > 2 ? uni(B3) : nullptr};
  auto issue = [&](int g, int slot) {
    const int pn = g / GPP, kq = (PROBE & 2) ? 0 : g % GPP;  // (probe 2: every group re-reads group 0 -- L1-resident operands)
    const gchar *a = (const gchar *)(Ap[pn] + wr * NB + kq * GK);
    const gchar *b = (const gchar *)(Bp[pn] + wc * NB + kq * GK);
#pragma unroll
    for (int m = 0; m < 4; m++) {
      ra[slot][m] = *(const gvu *)(a + voff[m]);
      if (!(PROBE & 1)) rb[slot][m] = *(const gvu *)(b + voff[m]);  // (probe 1: half the loads, B := A)
      else rb[slot][m] = ra[slot][m];
    }
  };
  issue(0, 0);
  issue(1, 1);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if ((PROBE & 4) || ((PROBE & 8) && (g & 3) == 0) || ((PROBE & 16) && (g & 1) == 0)) __builtin_amdgcn_s_barrier();  // (probe: keep the four waves in step so that the slices two of them share hit in L1)
    if (g + 2 < NG) issue(g + 2, (g + 2) % R);
#pragma unroll
    for (int s = 0; s < VL; s++)
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = RT<T>::mfma(ra[g % R][m][s], rb[g % R][n][s], acc[m][n]);
  }
}

constexpr int DKC = 8;
constexpr size_t GEMM_DMA_LDS_ELEMS = (size_t)4 * 2 * 2 * 64 * DKC;
__device__ inline void tile_gemm_abt_dma(const double *__restrict__ A0, const double *__restrict__ B0,
                                         const double *__restrict__ A1, const double *__restrict__ B1, double *lds,
                                         RT<double>::v4 acc[4][4]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  const int fr = lane & 15, fk = lane >> 4;
  double *wbase = lds + wv * (2 * 2 * 64 * DKC);  // [buffer][A | B][64 rows][8]
  const int prow = lane >> 2, pc2 = lane & 3;     // a piece: 16 rows x 64 B; this lane's row and 16-byte slot inside it
  int off[2][4];  // operand read offsets inside a 64 x 8 slice: row r, element k -> r*8 + (((k>>1) ^ ((r>>2)&3)) << 1) + (k&1)
#pragma unroll
  for (int kk = 0; kk < 2; kk++)
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int r = 16 * m + fr, k = 4 * kk + fk;
      off[kk][m] = r * DKC + (((k >> 1) ^ ((r >> 2) & 3)) << 1) + (k & 1);
    }
  constexpr int NCHK = 2 * NB / DKC, HALF = NB / DKC;
  auto issue = [&](int ch) {  // 8 DMA pieces: chunk ch of A and B into buffer ch & 1
    double *dst = wbase + (ch & 1) * (2 * 64 * DKC);
    const double *A = (ch < HALF ? A0 : A1) + (size_t)wr * NB, *B = (ch < HALF ? B0 : B1) + (size_t)wc * NB;
    const int k0 = (ch & (HALF - 1)) * DKC;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 16 * q + prow;
      const int c2 = pc2 ^ ((r >> 2) & 3);
      __builtin_amdgcn_global_load_lds(A + (size_t)r * NB + k0 + 2 * c2, dst + q * 16 * DKC, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(B + (size_t)r * NB + k0 + 2 * c2, dst + 64 * DKC + q * 16 * DKC, 16, 0, 0);
    }
  };
  issue(0);
  for (int ch = 0; ch < NCHK; ch++) {
    if (ch + 1 < NCHK) {
      issue(ch + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // chunk ch has landed; chunk ch + 1 (8 pieces) may still be in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const double *cA = wbase + (ch & 1) * (2 * 64 * DKC), *cB = cA + 64 * DKC;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      double af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; m++) af[m] = cA[off[kk][m]];
#pragma unroll
      for (int n = 0; n < 4; n++) bf[n] = cB[off[kk][n]];
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = RT<double>::mfma(af[m], bf[n], acc[m][n]);
    }
  }
}


template <typename T, int MODE, int DBG, bool OWN = false>
__global__ __launch_bounds__(256, 2) void k_ldl_update_probe(T *__restrict__ S, const int64_t *__restrict__ co, const T *__restrict__ V0,
                                                        const T *__restrict__ V1, int k, int base, int nt,
                                                        int nblk, int *__restrict__ ready,
                                                        const int *__restrict__ own_cols = nullptr,
                                                        const int64_t *__restrict__ own_pref = nullptr, int m0 = 0,
                                                        int m_end = 0, int ready_tiles = 1, const int *__restrict__ rows = nullptr,
                                                        const T *__restrict__ Lp0 = nullptr, const T *__restrict__ Lp1 = nullptr) {
  BA_VT
  static_assert(MODE == 1, "only the pair update is a tile-per-workgroup kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T *lds = reinterpret_cast<T *>(smraw);
  T *sA = lds, *sB = lds + NB * LDK;
  int i, j, tsel;
  {
    // chunked block -> XCD map: blocks b, b+8, ... share an XCD (round-robin dispatch); give each XCD a contiguous
    // range of tile rows so that V_i stays in its L2 (speed only)
    const int per = (nblk + 7) / 8;
    int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= nblk) return;
    tsel = t;
    if (OWN) {
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
      int ii = (int)((sqrt(8.0 * (T)t + 1.0) - 1.0) * 0.5);
      while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
      while (ii * (ii + 1) / 2 > t) ii--;
      int jj = t - ii * (ii + 1) / 2;
      if constexpr ((DBG & 8192) != 0) tri_blocked(t, nt - base, &ii, &jj);  // super-block enumeration (L2 reuse of both operands)
      // rows (block-sparse S): the pair's pattern, ascending -- tile (rows[ii], rows[jj]) instead of (base + ii, base + jj)
      i = rows ? rows[ii] : base + ii;
      j = rows ? rows[jj] : base + jj;
    }
  }
  T *Sij = S + tix(co, i, j) * NB * NB;
  typename RT<T>::v4 acc[4][4];
  if constexpr ((DBG & 64) != 0) {
    // probe: the accumulators start as -C (loads in flight beside the first operand chunk), the epilogue stores -acc
    const int lane0 = threadIdx.x & 63, wv0 = threadIdx.x >> 6;
    const T *c0 = Sij + ((wv0 >> 1) * 64) * NB + (wv0 & 1) * 64 + (lane0 & 15);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < 4; n++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[m][n][g] = -c0[(16 * m + RT<T>::row(lane0, g)) * NB + 16 * n];
  } else {
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < 4; n++) acc[m][n] = (d4){0, 0, 0, 0};
  }
  const int io = (DBG & 2) ? base : i, jo = (DBG & 2) ? base : j;
  if constexpr ((DBG & 16) != 0 && sizeof(T) == 8)
    tile_gemm_abt_dma(V0 + (int64_t)io * NB * NB, S + tix(co, jo, k) * NB * NB, V1 + (int64_t)io * NB * NB,
                      S + tix(co, jo, k + 1) * NB * NB, lds, acc);
  else if constexpr ((DBG & 32) != 0)  // K = 512 probe (tools/bench_update.py variant 32): panels k .. k+3, V2 / V3 behind V1
    tile_gemm_abt_priv<T, 4>(V0 + (int64_t)io * NB * NB, S + tix(co, jo, k) * NB * NB, V1 + (int64_t)io * NB * NB,
                          S + tix(co, jo, k + 1) * NB * NB, lds, acc, V1 + (int64_t)(nt + io) * NB * NB,
                          S + tix(co, jo, k + 2) * NB * NB, V1 + (int64_t)(2 * nt + io) * NB * NB, S + tix(co, jo, k + 3) * NB * NB);
  else if constexpr ((DBG & 128) != 0)  // direct operand loads, no LDS (bits 8 / 9: half the loads / L1-resident operands)
    tile_gemm_abt_direct<T, 2, ((DBG & 256) ? 1 : 0) | ((DBG & 512) ? 2 : 0) | ((DBG & 1024) ? 4 : 0) | ((DBG & 2048) ? 8 : 0) | ((DBG & 4096) ? 16 : 0)>(V0 + (int64_t)io * NB * NB, S + tix(co, jo, k) * NB * NB, V1 + (int64_t)io * NB * NB,
                               S + tix(co, jo, k + 1) * NB * NB, acc);
  else if (!(DBG & 8))
    // Lp0 / Lp1 (distributed factorisation with per-rank ownership of S): the L tiles of the two panels come from the panel
    // buffers the broadcast filled (tile row j at Lp + j NB^2) -- a rank holds only its own tile columns of S
    tile_gemm_abt_priv<T, 2>(V0 + (int64_t)io * NB * NB, Lp0 ? Lp0 + (int64_t)jo * NB * NB : S + tix(co, jo, k) * NB * NB,
                          V1 + (int64_t)io * NB * NB, Lp1 ? Lp1 + (int64_t)jo * NB * NB : S + tix(co, jo, k + 1) * NB * NB, lds, acc);
  else
    tile_gemm_abt<T, 2>(V0 + (int64_t)io * NB * NB, S + tix(co, jo, k) * NB * NB, V1 + (int64_t)io * NB * NB,
                     S + tix(co, jo, k + 1) * NB * NB, sA, sB, acc);
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));  // keep the epilogue's address arithmetic out of the main loop's live ranges
  const int lane = tid2 & 63, wv = tid2 >> 6;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;
  // epilogue: the 64 values of a lane are read-modify-written in two batches of 32 so that 32 loads are in flight at once
  T *cbase = Sij + wr * NB + wc + (lane & 15);
  if constexpr ((DBG & 64) != 0) {
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int g = 0; g < 4; g++) cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * n] = -acc[m][n][g];
  } else
#pragma unroll
  for (int h = 0; h < 2; h++) {
    T cv[2][4][4];
    if (!(DBG & 1)) {
#pragma unroll
      for (int n2 = 0; n2 < 2; n2++)
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int g = 0; g < 4; g++) cv[n2][m][g] = NT_C ? __builtin_nontemporal_load(&cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)]) : cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)];
    }
#pragma unroll
    for (int n2 = 0; n2 < 2; n2++)
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const T a = acc[m][2 * h + n2][g];
          const T nv = (DBG & 1) ? a : cv[n2][m][g] - a;
          if (NT_C) __builtin_nontemporal_store(nv, &cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)]);
          else cbase[(16 * m + RT<T>::row(lane, g)) * NB + 16 * (2 * h + n2)] = nv;
        }
  }
  // tiles 0 .. ready_tiles-1 are (base,base) [, (base+1,base), (base+1,base+1)]: what the next pair's hoisted diagonal
  // kernel waits for.  Each tells it so once its tile is final.
  if (ready && tsel < ready_tiles) {
    __threadfence();  // every thread's stores, agent scope (written back past this XCD's L2)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

namespace {
__global__ void k_probe_fill(double *a, size_t n, double scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (scale < 0) {  // full-entropy mantissas (splitmix64), values in (-|scale|, |scale|)
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    a[i] = -scale * ((double)(long long)z / 9223372036854775808.0);
  } else {
    a[i] = scale * (double)((i * 2654435761u) % 1021) / 1021.0 - 0.3;
  }
}
}  // namespace

static int set_bench_kernel_attrs() {
  typedef double T;
  BA_CHECK(set_kernel_attrs<T>());
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 8192>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 8193>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 3>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 17>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 32>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 33>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_PRIV_LDS_ELEMS * sizeof(T))));
  BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ldl_update_probe<T, 1, 9>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GEMM_LDS_ELEMS * sizeof(T))));
  return BA_OK;
}

// micro-benchmark of the bulk trailing update (tools/bench_update.py): one pair update of an nt x nt tile matrix
extern "C" int ba_debug_update_bench(int nt, int variant, int reps, double *ms_out) {
  BA_CHECK(set_bench_kernel_attrs());
  const size_t tiles = (size_t)nt * (nt + 1) / 2 * NB * NB;
  double *S = nullptr, *V = nullptr;
  BA_HIP_CHECK(hipMalloc((void **)&S, tiles * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&V, (size_t)4 * nt * NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMemset(S, 0, tiles * sizeof(double)));
  BA_HIP_CHECK(hipMemset(V, 0, (size_t)4 * nt * NB * NB * sizeof(double)));
  if (getenv("BA_BENCH_NONZERO")) {  // operands with real bit patterns (power / clocks differ from all-zero data)
    hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, 0, S, tiles, 1e-3);
    const double sc = getenv("BA_BENCH_ENTROPY") ? -1e-3 : 1e-3;  // negative: full-entropy mantissas
    hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, 0, S, tiles, sc);
    hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)(((size_t)4 * nt * NB * NB + 255) / 256)), dim3(256), 0, 0, V, (size_t)4 * nt * NB * NB, sc);
  }
  const bool quad = (variant & 32) != 0;  // K = 512: panels 0 .. 3, trailing matrix from tile column 4
  const int m = nt - (quad ? 4 : 2), nblk = m * (m + 1) / 2, grid = ((nblk + 7) / 8) * 8;
  std::vector<int64_t> h_co;
  dense_ldl_layout(nt, 1, &h_co, nullptr);
  int64_t *co_alloc = nullptr, *co = nullptr;  // (tix reads the table's head one entry before its pointer: 0 = dense)
  h_co.insert(h_co.begin(), 0);
  BA_HIP_CHECK(hipMalloc((void **)&co_alloc, h_co.size() * sizeof(int64_t)));
  BA_HIP_CHECK(hipMemcpy(co_alloc, h_co.data(), h_co.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  co = co_alloc + 1;
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  auto launch = [&]() {
    const double *V0 = V, *V1 = V + (size_t)nt * NB * NB;
    switch (variant) {
      case 1: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 1>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 2: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 2>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 8: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 8>), dim3(grid), dim3(256), GEMM_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 16: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 16>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 17: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 17>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 32: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 32>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 4, nt, nblk, (int *)nullptr); break;
      case 33: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 33>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 4, nt, nblk, (int *)nullptr); break;
      case 64: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 64>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 128: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 128>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 385: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 385>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 641: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 641>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 897: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 897>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 1152: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 1152>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 1153: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 1153>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 2176: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 2176>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 2177: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 2177>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 4224: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 4224>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 4225: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 4225>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 130: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 130>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 131: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 131>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 3: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 3>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 8192: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 8192>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 8193: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 8193>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 8320: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 8320>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 8321: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 8321>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 129: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 129>), dim3(grid), dim3(256), 0, 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      case 9: hipLaunchKernelGGL((k_ldl_update_probe<double, 1, 9>), dim3(grid), dim3(256), GEMM_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr); break;
      default: hipLaunchKernelGGL((k_ldl_update<double, 1>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0, 2, nt, nblk, (int *)nullptr);
    }
  };
  {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k_ldl_update<double, 1>), 256, GEMM_LDS_ELEMS * sizeof(double));
    if (variant == 0) fprintf(stderr, "[debug] update<1>: occupancy API says %d workgroups/CU at %zu B LDS\n", nb, GEMM_PRIV_LDS_ELEMS * sizeof(double));
  }
  launch();
  BA_HIP_CHECK(hipDeviceSynchronize());
  if (getenv("BA_BENCH_PERREP")) {  // every launch timed by its own event pair: drift of the rate under sustained load
    std::vector<hipEvent_t> ev((size_t)reps + 1);
    for (auto &e : ev) BA_HIP_CHECK(hipEventCreate(&e));
    BA_HIP_CHECK(hipEventRecord(ev[0], 0));
    for (int r = 0; r < reps; r++) {
      launch();
      BA_HIP_CHECK(hipEventRecord(ev[(size_t)r + 1], 0));
    }
    BA_HIP_CHECK(hipEventSynchronize(ev[(size_t)reps]));
    fprintf(stderr, "[perrep]");
    for (int r = 0; r < reps; r++) {
      float t = 0;
      BA_HIP_CHECK(hipEventElapsedTime(&t, ev[(size_t)r], ev[(size_t)r + 1]));
      fprintf(stderr, " %.3f", t);
    }
    fprintf(stderr, "\n");
    for (auto &e : ev) (void)hipEventDestroy(e);
  }
  BA_HIP_CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; r++) launch();
  BA_HIP_CHECK(hipEventRecord(e1, 0));
  BA_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / reps;
  (void)hipFree(S);
  (void)hipFree(V);
  (void)hipFree(co_alloc);
  return BA_OK;
}

// a sequence of pair updates of m_list[q] tile rows each on one nt x nt tile matrix (nonzero operands), every launch
// timed by its own event pair; gap_us[q] > 0: the host sleeps that long BEFORE launch q (tools/bench_update_seq.py)
extern "C" int ba_debug_update_seq(int nt, int n, const int *m_list, const int *gap_us, double *ms_out) {
  BA_CHECK(set_bench_kernel_attrs());
  const size_t tiles = (size_t)nt * (nt + 1) / 2 * NB * NB, vel = (size_t)2 * nt * NB * NB;
  double *S = nullptr, *V = nullptr;
  BA_HIP_CHECK(hipMalloc((void **)&S, tiles * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&V, vel * sizeof(double)));
  const double fs = getenv("BA_BENCH_RANDOM") ? -1.0 : 1e-3;  // random mantissas (what a real matrix has) or 1021 distinct values
  hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, 0, S, tiles, fs);
  hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((vel + 255) / 256)), dim3(256), 0, 0, V, vel, fs);
  std::vector<int64_t> h_co;
  dense_ldl_layout(nt, 1, &h_co, nullptr);
  int64_t *co_alloc = nullptr, *co = nullptr;  // (tix reads the table's head one entry before its pointer: 0 = dense)
  h_co.insert(h_co.begin(), 0);
  BA_HIP_CHECK(hipMalloc((void **)&co_alloc, h_co.size() * sizeof(int64_t)));
  BA_HIP_CHECK(hipMemcpy(co_alloc, h_co.data(), h_co.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  co = co_alloc + 1;
  std::vector<hipEvent_t> ev((size_t)2 * n);
  for (auto &e : ev) BA_HIP_CHECK(hipEventCreate(&e));
  BA_HIP_CHECK(hipDeviceSynchronize());
  const double *V0 = V, *V1 = V + (size_t)nt * NB * NB;
  for (int q = 0; q < n; q++) {
    if (gap_us && gap_us[q] > 0) {
      BA_HIP_CHECK(hipDeviceSynchronize());
      timespec ts = {0, (long)gap_us[q] * 1000};
      nanosleep(&ts, nullptr);
    }
    if (m_list[q] < 0) {  // -c: c passes of a memory-bound kernel over S (what precedes a factorisation in the LM loop)
      BA_HIP_CHECK(hipEventRecord(ev[(size_t)2 * q], 0));
      for (int c = 0; c < -m_list[q]; c++)
        hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, 0, S, tiles, 1e-3);
      BA_HIP_CHECK(hipEventRecord(ev[(size_t)2 * q + 1], 0));
      continue;
    }
    const int m = m_list[q], nblk = m * (m + 1) / 2, grid = ((nblk + 7) / 8) * 8;
    BA_HIP_CHECK(hipEventRecord(ev[(size_t)2 * q], 0));
    hipLaunchKernelGGL((k_ldl_update<double, 1>), dim3(grid), dim3(256), GEMM_PRIV_LDS_ELEMS * sizeof(double), 0, S, co, V0, V1, 0,
                       nt - m, nt, nblk, (int *)nullptr);
    BA_HIP_CHECK(hipEventRecord(ev[(size_t)2 * q + 1], 0));
  }
  BA_HIP_CHECK(hipDeviceSynchronize());
  for (int q = 0; q < n; q++) {
    float t = 0;
    BA_HIP_CHECK(hipEventElapsedTime(&t, ev[(size_t)2 * q], ev[(size_t)2 * q + 1]));
    ms_out[q] = t;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  (void)hipFree(S);
  (void)hipFree(V);
  (void)hipFree(co_alloc);
  return BA_OK;
}

// diagnostic: phase cycle counts of the diagonal-tile kernel (load, pivots, inverse16, trsm16+syrk16, full inverse, store)
extern "C" int ba_debug_diag_stamps(double *cycles6, double *ms_out) {
  BA_CHECK(set_bench_kernel_attrs());
  double *S = nullptr, *Li = nullptr, *D = nullptr;
  int *flag = nullptr;
  unsigned long long *st = nullptr;
  BA_HIP_CHECK(hipMalloc((void **)&S, NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&Li, NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMemset(Li, 0, NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&D, NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&flag, sizeof(int)));
  BA_HIP_CHECK(hipMalloc((void **)&st, 118 * sizeof(unsigned long long)));
  BA_HIP_CHECK(hipMemset(st, 0, 118 * sizeof(unsigned long long)));
  std::vector<double> h((size_t)NB * NB, 0.0);
  for (int i = 0; i < NB; i++)
    for (int j = 0; j <= i; j++) h[(size_t)i * NB + j] = (i == j) ? 300.0 + i : 1.0 / (1 + i + j);
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    BA_HIP_CHECK(hipMemcpy(S, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    BA_HIP_CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_ldl_diag<double>, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_ELEMS * sizeof(double), 0, S, Li, D, flag, rep == 2 ? st : nullptr, (const int *)nullptr);
    BA_HIP_CHECK(hipEventRecord(e1, 0));
    BA_HIP_CHECK(hipEventSynchronize(e1));
    if (rep == 1) BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  {  // fingerprint of the outputs (inverse tile + pivots): refactorings of the kernel are expected to keep every bit
    std::vector<double> li((size_t)NB * NB), dv(NB);
    BA_HIP_CHECK(hipMemcpy(li.data(), Li, li.size() * sizeof(double), hipMemcpyDeviceToHost));
    BA_HIP_CHECK(hipMemcpy(dv.data(), D, dv.size() * sizeof(double), hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    auto mix = [&](const std::vector<double> &v) {
      const unsigned char *b = reinterpret_cast<const unsigned char *>(v.data());
      for (size_t q = 0; q < v.size() * sizeof(double); q++) hsh = (hsh ^ b[q]) * 1099511628211ull;
    };
    mix(li);
    mix(dv);
    fprintf(stderr, "[diag] output fingerprint %016llx  D[0] %.17g D[127] %.17g Linv[127][0] %.17g\n", hsh, dv[0], dv[127], li[(size_t)127 * NB]);
  }
  unsigned long long hs[118];
  BA_HIP_CHECK(hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost));
  for (int q = 0; q < 118; q++) cycles6[q] = (double)hs[q];  // 6 phases, then the busy time of wave w in phase P(s) at [6 + 8 s + w] and in phase A(s+1) at [62 + 8 s + w]
  *ms_out = ms;
  (void)hipFree(S); (void)hipFree(Li); (void)hipFree(D); (void)hipFree(flag); (void)hipFree(st);
  return BA_OK;
}

// ---- MFMA ceiling probes (tools/bench_mfma_probe.py): how close a K loop of the update kernel's shape can get to the
// f64 matrix peak.  MODE 0: MFMAs only, 4x4 accumulator blocks per wave, 2 waves per SIMD; MODE 1: the same with the
// update kernel's LDS operand reads (8 ds_read_b64 per 16 MFMAs), no staging; MODE 2: 4x8 blocks per wave, one wave per
// SIMD, no staging.  Modes 3 / 4 (k_stage_probe): the full wave-private staging loop (global -> registers -> LDS, chunks
// of 16, next chunk's loads in flight during the MFMAs) on operands in global memory, K = 256 per "tile", 16 tiles per
// persistent workgroup, no C tile traffic: 3 = the shipped geometry (64x64 per wave, 2 workgroups per CU), 4 = 64x128
// per wave, 1 workgroup per CU.
// Measured (TFLOP/s): 0: 78.0   1: 78.0   2: 58.6 (compiler spills)   3: 67.7-69.8   4: 64.3-65.1.
// (Also tried in this probe: two LDS buffers of 8-wide chunks per wave at the same 73.7 KB, row stride 9 doubles: 43 --
// the 72-byte rows break the 16-byte LDS writes.)
// With them: the shipped kernel's 59.8 in tools/bench_update.py becomes 64-65 with a store-only epilogue (variant 1), and
// reading the C tile costs the same 6-7 TFLOP/s wherever the loads are placed (end of tile, or start of tile into the
// accumulators, with or without persistent workgroups): the pair update moves 2 x 0.98 GB of C per launch for 0.063
// TFLOP, i.e. at K = 256 the trailing matrix's HBM traffic is a third of the kernel's time when it is not overlapped.
namespace {
template <int MODE>
__global__ __launch_bounds__(256, (MODE == 2 ? 1 : 2)) void k_mfma_probe(double *out, int iters) {
  typedef double d4p __attribute__((ext_vector_type(4)));
  constexpr int NC = (MODE == 2) ? 8 : 4;
  __shared__ double sm[(64 + 128) * 18];
  for (int i = threadIdx.x; i < (64 + 128) * 18; i += 256) sm[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
  d4p acc[4][NC];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < NC; n++) acc[m][n] = (d4p){0, 0, 0, 0};
  double af[4], bf[NC];
#pragma unroll
  for (int m = 0; m < 4; m++) af[m] = 1.0 + lane * 1e-3 + m;
#pragma unroll
  for (int n = 0; n < NC; n++) bf[n] = 2.0 - lane * 1e-3 + n;
  const double *sA = sm, *sB = sm + 64 * 18;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      if (MODE >= 1) {
#pragma unroll
        for (int m = 0; m < 4; m++) af[m] = sA[(16 * m + fr) * 18 + kk * 4 + fk];
#pragma unroll
        for (int n = 0; n < NC; n++) bf[n] = sB[(((wv & 1) * 64 + 16 * n + fr) & 127) * 18 + kk * 4 + fk];
      }
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < NC; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < NC; n++) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// staging probe: NCB = column blocks of 16 per wave (4: 64x64 wave tile, 8: 64x128)
template <int NCB>
__global__ __launch_bounds__(256, (NCB == 8 ? 1 : 2)) void k_stage_probe(const double *__restrict__ Aop,
                                                                          const double *__restrict__ Bop, double *out,
                                                                          int tiles_per_wg, int ntile_rows) {
  typedef double d4p __attribute__((ext_vector_type(4)));
  typedef double d2p __attribute__((ext_vector_type(2)));
  constexpr int BR = 16 * NCB;  // B rows per wave
  constexpr int LD = 18;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double *lds = reinterpret_cast<double *>(smraw);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
  double *sA = lds + wv * ((64 + BR) * LD), *sB = sA + 64 * LD;
  const int lrow = lane >> 3, lc2 = lane & 7;  // 8 lanes cover the 16 doubles of a chunk row
  constexpr int NLA = 64 / 8, NLB = BR / 8;
  d4p acc[4][NCB];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < NCB; n++) acc[m][n] = (d4p){0, 0, 0, 0};
  for (int t = 0; t < tiles_per_wg; t++) {
    // operand tiles of this step: row-major tiles of 128 x 128 doubles; every workgroup walks its own sequence
    const int ti = (blockIdx.x * 7 + t * 3) % ntile_rows, tj = (blockIdx.x * 5 + t) % ntile_rows;
    const double *Ab = Aop + (size_t)ti * NB * NB + (size_t)((wv >> 1) * 64) * NB;
    const double *Bb = Bop + (size_t)tj * NB * NB + (size_t)(((wv & 1) * BR) % NB) * NB;
    d2p pa[NLA], pb[NLB];
#pragma unroll
    for (int q = 0; q < NLA; q++) pa[q] = *reinterpret_cast<const d2p *>(Ab + (lrow + 8 * q) * NB + 2 * lc2);
#pragma unroll
    for (int q = 0; q < NLB; q++) pb[q] = *reinterpret_cast<const d2p *>(Bb + ((lrow + 8 * q) % NB) * NB + 2 * lc2);
    for (int ch = 0; ch < 16; ch++) {  // K = 256 in chunks of 16
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NLA; q++) *reinterpret_cast<d2p *>(sA + (lrow + 8 * q) * LD + 2 * lc2) = pa[q];
#pragma unroll
      for (int q = 0; q < NLB; q++) *reinterpret_cast<d2p *>(sB + (lrow + 8 * q) * LD + 2 * lc2) = pb[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (ch + 1 < 16) {
        const int k0 = ((ch + 1) & 7) * 16;
#pragma unroll
        for (int q = 0; q < NLA; q++) pa[q] = *reinterpret_cast<const d2p *>(Ab + (lrow + 8 * q) * NB + k0 + 2 * lc2);
#pragma unroll
        for (int q = 0; q < NLB; q++) pb[q] = *reinterpret_cast<const d2p *>(Bb + ((lrow + 8 * q) % NB) * NB + k0 + 2 * lc2);
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        double af[4], bf[NCB];
#pragma unroll
        for (int m = 0; m < 4; m++) af[m] = sA[(16 * m + fr) * LD + kk * 4 + fk];
#pragma unroll
        for (int n = 0; n < NCB; n++) bf[n] = sB[(16 * n + fr) * LD + kk * 4 + fk];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int n = 0; n < NCB; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  double s = 0;  // consume the accumulators one column block at a time (no 256-register reduction)
#pragma unroll
  for (int n = 0; n < NCB; n++) {
    double sn = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) sn += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    s += sn;
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
// Mode 5: the same wave-private loop with LDS-DMA staging (global_load_lds_dwordx4: no staging VGPRs, no ds_write): chunks of 8,
// two LDS buffers per wave (16 KB per wave, 64 KB per workgroup), the next chunk's 8 DMA pieces in flight during the MFMAs, the
// wave waits with vmcnt only -- no barrier.  A DMA piece lands at base + lane * 16: rows are unpadded (64 B), so the 16-byte
// columns are XOR-swizzled with the row ((row >> 2) & 3) to keep the MFMA operand reads at two dwords per bank.
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void k_dma_probe(const double *__restrict__ Aop, const double *__restrict__ Bop, double *out,
                                                      int tiles_per_wg, int ntile_rows) {
  typedef double d4p __attribute__((ext_vector_type(4)));
  constexpr int DKC = 8, NCHK = 256 / DKC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double *lds = reinterpret_cast<double *>(smraw);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
  double *wbase = lds + wv * (2 * 2 * 64 * DKC);  // [buffer][A | B][64 rows][8]
  const int prow = lane >> 2, pc2 = lane & 3;     // a piece: 16 rows x 64 B; this lane's row and 16-byte slot inside it
  d4p acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) acc[m][n] = (d4p){0, 0, 0, 0};
  // operand read offsets (doubles) inside a 64 x 8 slice: row r, element k -> r * 8 + (((k >> 1) ^ ((r >> 2) & 3)) << 1) + (k & 1)
  int offA[2][4], offB[2][4];
#pragma unroll
  for (int kk = 0; kk < 2; kk++)
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int r = 16 * m + fr, k = 4 * kk + fk;
      offA[kk][m] = r * DKC + ((((k >> 1) ^ ((r >> 2) & 3))) << 1) + (k & 1);
      offB[kk][m] = offA[kk][m];
    }
  for (int t = 0; t < tiles_per_wg; t++) {
    const int ti = (blockIdx.x * 7 + t * 3) % ntile_rows, tj = (blockIdx.x * 5 + t) % ntile_rows;
    const double *Ab = Aop + (size_t)ti * NB * NB + (size_t)((wv >> 1) * 64) * NB;
    const double *Bb = Bop + (size_t)tj * NB * NB + (size_t)((wv & 1) * 64) * NB;
    auto issue = [&](int ch) {  // 8 DMA pieces: chunk ch of A and B into buffer ch & 1
      double *dst = wbase + (ch & 1) * (2 * 64 * DKC);
      const int k0 = (ch & (128 / DKC - 1)) * DKC;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 16 * q + prow;
        const int c2 = pc2 ^ ((r >> 2) & 3);
        __builtin_amdgcn_global_load_lds(Ab + (size_t)r * NB + k0 + 2 * c2, dst + q * 16 * DKC, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(Bb + (size_t)r * NB + k0 + 2 * c2, dst + 64 * DKC + q * 16 * DKC, 16, 0, 0);
      }
    };
    issue(0);
    for (int ch = 0; ch < NCHK; ch++) {
      if (ch + 1 < NCHK) {
        issue(ch + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // chunk ch has landed, chunk ch + 1 (8 pieces) may still be in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const double *cA = wbase + (ch & 1) * (2 * 64 * DKC), *cB = cA + 64 * DKC;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        double af[4], bf[4];
#pragma unroll
        for (int m = 0; m < 4; m++) af[m] = cA[offA[kk][m]];
#pragma unroll
        for (int n = 0; n < 4; n++) bf[n] = cB[offB[kk][n]];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    double sn = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) sn += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    s += sn;
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

// The diagonal-tile kernel (one workgroup, a chain of dependent instructions) alone on the device against the same kernel
// beside `fill_blocks` workgroups of pure MFMA work on another stream: is its time set by the clock the device grants an
// almost idle chip?  us_out: n launches alone, then n launches started right behind the filler (tools/bench_diag_busy.py)
extern "C" int ba_debug_diag_busy(int n, int fill_blocks, int fill_iters, double *us_out) {
  BA_CHECK(set_bench_kernel_attrs());
  double *S = nullptr, *Li = nullptr, *D = nullptr, *out = nullptr;
  int *flag = nullptr;
  BA_HIP_CHECK(hipMalloc((void **)&S, NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&Li, NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMemset(Li, 0, NB * NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&D, NB * sizeof(double)));
  BA_HIP_CHECK(hipMalloc((void **)&flag, sizeof(int)));
  BA_HIP_CHECK(hipMalloc((void **)&out, (size_t)(fill_blocks > 0 ? fill_blocks : 1) * 256 * sizeof(double)));
  std::vector<double> h((size_t)NB * NB, 0.0);
  for (int i = 0; i < NB; i++)
    for (int j = 0; j <= i; j++) h[(size_t)i * NB + j] = (i == j) ? 300.0 + i : 1.0 / (1 + i + j);
  BA_HIP_CHECK(hipMemcpy(S, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  BA_HIP_CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  BA_HIP_CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev((size_t)2 * n + 2);
  for (auto &e : ev) BA_HIP_CHECK(hipEventCreate(&e));
  for (int phase = 0; phase < 2; phase++) {
    BA_HIP_CHECK(hipDeviceSynchronize());
    if (phase == 1 && fill_blocks > 0)
      hipLaunchKernelGGL(k_mfma_probe<0>, dim3(fill_blocks), dim3(256), 0, sb, out, fill_iters);
    BA_HIP_CHECK(hipEventRecord(ev[0], sa));
    for (int q = 0; q < n; q++) {  // (the kernel only reads S: every launch factors the same tile)
      hipLaunchKernelGGL(k_ldl_diag<double>, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_ELEMS * sizeof(double), sa, S, Li, D, flag,
                         (unsigned long long *)nullptr, (const int *)nullptr);
      BA_HIP_CHECK(hipEventRecord(ev[(size_t)q + 1], sa));
    }
    BA_HIP_CHECK(hipDeviceSynchronize());
    for (int q = 0; q < n; q++) {
      float t = 0;
      BA_HIP_CHECK(hipEventElapsedTime(&t, ev[(size_t)q], ev[(size_t)q + 1]));
      us_out[phase * n + q] = 1e3 * t;
    }
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(sa); (void)hipStreamDestroy(sb);
  (void)hipFree(S); (void)hipFree(Li); (void)hipFree(D); (void)hipFree(flag); (void)hipFree(out);
  return BA_OK;
}

extern "C" int ba_debug_mfma_probe(int mode, int iters, double *tflops_out) {
  int dev = 0, ncu = 256;
  BA_HIP_CHECK(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  double *out = nullptr, *Aop = nullptr, *Bop = nullptr;
  float ms = 0;
  double flops = 0;
  if (mode <= 2) {
    const int per_cu = mode == 2 ? 1 : 2, grid = ncu * per_cu * 4;
    BA_HIP_CHECK(hipMalloc((void **)&out, (size_t)grid * 256 * sizeof(double)));
    auto launch = [&]() {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_mfma_probe<0>, dim3(grid), dim3(256), 0, 0, out, iters); break;
        case 1: hipLaunchKernelGGL(k_mfma_probe<1>, dim3(grid), dim3(256), 0, 0, out, iters); break;
        default: hipLaunchKernelGGL(k_mfma_probe<2>, dim3(grid), dim3(256), 0, 0, out, iters);
      }
    };
    launch();
    BA_HIP_CHECK(hipDeviceSynchronize());
    BA_HIP_CHECK(hipEventRecord(e0, 0));
    launch();
    BA_HIP_CHECK(hipEventRecord(e1, 0));
    BA_HIP_CHECK(hipEventSynchronize(e1));
    BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double nc = mode == 2 ? 8 : 4;
    flops = (double)grid * 4 /*waves*/ * iters * 4 /*kk*/ * 4 * nc * 2048.0;
  } else {
    const int ncb = mode == 4 ? 8 : 4, per_cu = mode == 4 ? 1 : 2, grid = ncu * per_cu, tiles = iters > 0 ? iters : 16;
    const int ntile_rows = 120;  // 2 x 15.7 MB of operands: cache-resident like the panels of a pair update
    const size_t lds_bytes = mode == 5 ? (size_t)4 * 2 * 2 * 64 * 8 * sizeof(double) : (size_t)4 * (64 + 16 * ncb) * 18 * sizeof(double);
    BA_HIP_CHECK(hipMalloc((void **)&out, (size_t)grid * 256 * sizeof(double)));
    BA_HIP_CHECK(hipMalloc((void **)&Aop, (size_t)ntile_rows * NB * NB * sizeof(double)));
    BA_HIP_CHECK(hipMalloc((void **)&Bop, (size_t)ntile_rows * NB * NB * sizeof(double)));
    {  // non-trivial operands: the checksum of out[] must be the same for every staging variant (same products, same order)
      const size_t ne = (size_t)ntile_rows * NB * NB;
      hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, 0, Aop, ne, 1.0);
      hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, 0, Bop, ne, 0.7);
    }
    if (mode == 5)
      BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dma_probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    else if (mode == 3)
      BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage_probe<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    else
      BA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage_probe<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    auto launch = [&]() {
      if (mode == 5) hipLaunchKernelGGL(k_dma_probe<0>, dim3(grid), dim3(256), lds_bytes, 0, Aop, Bop, out, tiles, ntile_rows);
      else if (mode == 3) hipLaunchKernelGGL(k_stage_probe<4>, dim3(grid), dim3(256), lds_bytes, 0, Aop, Bop, out, tiles, ntile_rows);
      else hipLaunchKernelGGL(k_stage_probe<8>, dim3(grid), dim3(256), lds_bytes, 0, Aop, Bop, out, tiles, ntile_rows);
    };
    launch();
    BA_HIP_CHECK(hipDeviceSynchronize());
    BA_HIP_CHECK(hipEventRecord(e0, 0));
    launch();
    BA_HIP_CHECK(hipEventRecord(e1, 0));
    BA_HIP_CHECK(hipEventSynchronize(e1));
    BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    flops = (double)grid * 4 * tiles * 16 /*chunks*/ * 4 /*kk*/ * 4 * ncb * 2048.0;
    {
      std::vector<double> h((size_t)grid * 256);
      BA_HIP_CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
      double cs = 0, ca = 0;
      for (double v : h) { cs += v; ca += std::fabs(v); }
      fprintf(stderr, "[probe] mode %d: checksum %.17g (abs %.17g) over %zu values\n", mode, cs, ca, h.size());
    }
  }
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  if (out) (void)hipFree(out);
  if (Aop) (void)hipFree(Aop);
  if (Bop) (void)hipFree(Bop);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return BA_OK;
}
