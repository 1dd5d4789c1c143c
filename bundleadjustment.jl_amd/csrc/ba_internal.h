// Internal declarations shared by the translation units of libba_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"
#include "ba_order.h"  // TilePattern, camera orderings (host only)

void ba_set_error(const char *fmt, ...);

#define BA_HIP_CHECK(expr)                                                                          \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      ba_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));            \
      return BA_ERR_HIP;                                                                            \
    }                                                                                               \
  } while (0)

#define BA_CHECK(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc != BA_OK) return _rc; \
  } while (0)

// ---- dense reduced-camera system ------------------------------------------------------------------
// S is stored as the lower block triangle of NB x NB tiles, each tile contiguous row-major (128 KiB).  Tile (i,j),
// j <= i, sits at tile index col_off[j] + (i - j): a tile column is contiguous (rows j..nt-1).  Tile columns are taken
// in PAIRS (2q, 2q+1) -- the unit the factorisation works in -- and the pairs are grouped by owner rank q mod world
// (ascending q inside a group), so that the part of S a rank owns in the distributed factorisation is ONE contiguous
// range (a single reduce onto its owner per rank).  With world == 1 this is plain column order.
constexpr int NB = 128;
// Tile (i, j), j <= i, of the packed reduced camera matrix sits at tile offset tix(col_off, i, j).  col_off points ONE PAST the
// head of its table: col_off[-1] = 0 for the dense layout (a tile column is contiguous: col_off[j] + (i - j)), or = nt for the
// compressed layout of a block-sparse pattern, where col_off[nt + i nt + j] is the position of tile row i among the stored
// rows of column j (negative, and so a negative result, for a tile outside the pattern: nothing is allocated for it).
// Enumeration of the lower triangle of an m x m tile grid in SUPER-BLOCKS of TSB x TSB tiles (block rows top to bottom, blocks
// left to right, row-major inside a block; the diagonal blocks hold their lower triangle): t -> (ii, jj), jj <= ii.  The
// workgroups of the bulk trailing update that run at the same time (consecutive t) then share TSB row panels of each
// operand instead of one panel of A and one panel PER TILE of B -- the operand reads that miss the XCD's L2 drop by ~TSB / 2.
// t = 0, 1, 2 are (0,0), (1,0), (1,1): what the hoisted diagonal kernels wait for.
constexpr int TSB = 8;
__host__ __device__ inline void tri_blocked(int t, int m, int *ii, int *jj, int sb = TSB) {
  // tiles before block row bi (all block rows above it are full): sum_{r < bi} (sb^2 r + sb (sb + 1) / 2)
  const int full = sb * sb, tri = sb * (sb + 1) / 2;
  int lo = 0, hi = (m + sb - 1) / sb;  // largest bi with start(bi) <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (full * mid * (mid - 1) / 2 + tri * mid <= t) lo = mid;
    else hi = mid;
  }
  const int bi = lo;
  int rem = t - (full * bi * (bi - 1) / 2 + tri * bi);
  const int h = (m - sb * bi) < sb ? (m - sb * bi) : sb;  // tile rows in this block row
  if (rem < bi * sb * h) {
    const int bj = rem / (sb * h), r2 = rem - bj * (sb * h);
    *ii = sb * bi + r2 / sb;
    *jj = sb * bj + r2 % sb;
  } else {
    rem -= bi * sb * h;
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= rem) r++;
    *ii = sb * bi + r;
    *jj = sb * bi + rem - r * (r + 1) / 2;
  }
}

__host__ __device__ inline int64_t tix(const int64_t *__restrict__ col_off, int64_t i, int64_t j) {
  const int64_t n = col_off[-1];
  return col_off[j] + (n ? col_off[n + i * n + j] : (i - j));
}
constexpr int64_t BA_NO_TILE = -((int64_t)1 << 40);
// host: fill col_off (nt entries) and, when own_range != null, the [begin, end) tile ranges of the `world` owners
void dense_ldl_layout(int64_t nt, int world, std::vector<int64_t> *col_off, std::vector<int64_t> *own_range);

enum ProfClass {
  PC_RESIDUAL = 0,
  PC_JAC_STRUCTURE,
  PC_JAC_COORD,
  PC_POINT_BLOCKS,
  PC_CAM_BLOCKS,
  PC_SCHUR_PREP,
  PC_SCHUR_S,
  PC_SCHUR_RHS,
  PC_LDL_DIAG,
  PC_LDL_TRSM,
  PC_LDL_SYRK,    // update of the next tile column inside a panel pair (k_ldl_col_rs)
  PC_LDL_UPDATE,  // bulk pair update of the trailing matrix (k_ldl_update)
  PC_LDL_UPDATE_RS,  // short pair updates in row-split form (k_ldl_update_rs)
  PC_SOLVE,
  PC_BACKSUB,
  PC_TRIAL,
  PC_REDUCE,
  PC_COMM,
  PC_COUNT
};
extern const char *const kProfNames[PC_COUNT];

struct ProfSlot {
  double ms = 0;
  int64_t calls = 0;
};

template <typename T>
struct DenseLDLT {  // workspace of the blocked LDL^T in scalar type T, n = 9*ncams padded to nt*NB
  int64_t n = 0, nt = 0;
  T *S = nullptr;          // packed lower tiles (may alias the caller's reduce buffer)
  T *V = nullptr;          // 4 x nt tiles: V_i = L_ik * D_k of two panel pairs (double-buffered for the look-ahead)
  T *Linv = nullptr;       // nt tiles: inverse of each unit-lower diagonal tile
  T *D = nullptr;          // nt*NB pivots (+ nt*NB scratch)
  int64_t *col_off = nullptr;           // device: tile column offsets (see tix)
  std::vector<int64_t> h_col_tab;       // host copy of the table: [head | nt column offsets | (compressed) nt x nt row positions]
  int64_t *col_tab = nullptr;           // the device allocation (col_off = col_tab + 1)
  const int64_t *hco() const { return h_col_tab.data() + 1; }
  int64_t *hco() { return h_col_tab.data() + 1; }
  std::vector<int64_t> own_range;       // world + 1 tile offsets: rank r owns tiles [own_range[r], own_range[r+1])
  int world = 1, rank = 0;              // distribution of the tile column pairs (owner of pair q: q % world)
  std::vector<int> h_own_cols;          // tile columns owned by this rank, ascending
  std::vector<int64_t> h_own_pref;      // h_own_pref[m] = tiles in the owned columns before h_own_cols[m]
  int *own_cols = nullptr;              // device copies
  int64_t *own_pref = nullptr;
  double *flag_sum = nullptr;           // device double: the pivot flag on its way through the all-reduce
  int *flag = nullptr;     // device int: set to 1 on an exactly zero pivot (2: a hoisted diagonal tile never became ready)
  int *ready = nullptr;    // nt device ints: tile (k,k) has received its last trailing update (hoisted-diagonal schedule)
  hipStream_t hoist = nullptr;   // second stream of the hoisted-diagonal schedule (no CU mask)
  hipStream_t rest = nullptr;    // look-ahead of the block-sparse schedule: the rest of a pair's update (lowest priority)
  hipEvent_t ev_top = nullptr;
  bool hoist_disabled = false;   // a hoisted kernel once timed out (kernels serialised by a profiler): never again on this handle
  bool hoisting = false;         // the factorisation forks onto `hoist` (set by dense_ldl_factor's schedule choice)
  bool own_S = true;
  hipEvent_t ev_chain = nullptr;  // recorded behind each hoisted diagonal kernel
  // distributed factorisation with look-ahead: panels of pair q received (transfer stream), update of pair q launched
  hipEvent_t ev_recv[2] = {nullptr, nullptr}, ev_upd[2] = {nullptr, nullptr};
  // per-rank ownership of S (distributed factorisation): S holds this rank's tile columns only (s_tiles tiles), col_off /
  // hco() are rank-local offsets (negative for other ranks' columns), h_glob_off the owner-major global layout;
  // Lb: L = V D^-1 of the panel pairs in flight (the layout of V), bpart: partial products of the backward sweep
  bool own_only = false;
  int64_t s_tiles = 0;
  std::vector<int64_t> h_glob_off;
  T *Lb = nullptr, *bpart = nullptr;
  // block-sparse S (one GPU): the pattern's row / column lists on the device; null pattern = dense
  bool sparse = false;
  const TilePattern *pat = nullptr;
  int *prow = nullptr, *lcol = nullptr, *lpair = nullptr;
  // ... on several ranks (per-rank ownership of the PATTERN's tile columns): tiles stored per tile column, the (i, j) of every
  // tile this rank stores in storage order (column scaling), and per tile column pair q the tiles (i, j) -- both in U_q, j
  // owned by this rank, sorted by (j, i) -- its trailing update touches here; the first h_upd_lead[q] of them lie in the next
  // pair's own tile columns (look-ahead of the distributed factorisation)
  std::vector<int64_t> h_col_cnt;
  std::vector<int> h_upd_ptr, h_upd_lead;
  int2 *upd_ij = nullptr, *own_tiles = nullptr;
  hipEvent_t ev_dtop = nullptr, ev_dchain = nullptr;  // distributed factorisation: fork behind the reduce of S, end of the owner's panel chain
};
typedef DenseLDLT<double> DenseLDL;

// transport of the cross-rank sums (ba_comm.hip): RCCL called directly, or a caller-supplied hook
struct BaComm {
  int rank = 0, world = 1;
  ba_comm_fn hook = nullptr;
  void *hook_ctx = nullptr;
  void *nccl = nullptr;  // ncclComm_t
  int64_t calls = 0, bytes = 0;
  int64_t op_calls[BA_COMM_OPS] = {}, op_bytes[BA_COMM_OPS] = {};
  bool active() const { return hook != nullptr || nccl != nullptr; }  // a 1-rank communicator still exercises the path
};

struct ba_problem {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t ncams = 0, npnts = 0, nobs = 0;
  // device mirrors (0-based int32)
  int *cam0 = nullptr, *pnt0 = nullptr;
  double *pt2d = nullptr;
  float *pt2d_f32 = nullptr;
  // observation lists sorted by point / by camera (stable) for the deterministic reductions
  int *pt_ptr = nullptr, *pt_obs = nullptr;    // npnts+1, nobs
  int *cam_ptr = nullptr, *cam_obs = nullptr;  // ncams+1, nobs
  bool point_sorted = false;                   // observations already grouped by point (BAL order)
  std::vector<int> h_cam0, h_pnt0, h_pt_ptr, h_pt_obs;
  // scratch for the host-pointer entries
  void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes[4] = {0, 0, 0, 0};
  // LM workspace (allocated at the first solve)
  struct LMWork *lm = nullptr;
  // communication (multi-GPU)
  int rank = 0, world = 1;
  BaComm comm;
  // profiling
  bool prof_on = false;
  ProfSlot prof[PC_COUNT];
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// RAII-less helper: times one kernel class with an event pair when profiling is on (synchronises).
struct ProfScope {
  ba_problem *p;
  int cls;
  hipStream_t st;
  ProfScope(ba_problem *p_, int cls_, hipStream_t st_) : p(p_), cls(cls_), st(st_) {
    if (p->prof_on) (void)hipEventRecord(p->ev0, st);
  }
  ~ProfScope() {
    if (p->prof_on) {
      (void)hipEventRecord(p->ev1, st);
      (void)hipEventSynchronize(p->ev1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, p->ev0, p->ev1);
      p->prof[cls].ms += ms;
      p->prof[cls].calls += 1;
    }
  }
};

int ba_scratch(ba_problem *p, int slot, size_t bytes, void **out);

// ---- launchers (ba_model_kernels.hip) -----------------------------------------------------------
int launch_residual_f64(ba_problem *p, const double *d_x, double *d_r, hipStream_t st);
int launch_residual_f32(ba_problem *p, const float *d_x, float *d_r, hipStream_t st);
int launch_jac_structure(ba_problem *p, int64_t *d_rows, int64_t *d_cols, hipStream_t st);
int launch_jac_coord_f64(ba_problem *p, const double *d_x, double *d_vals, hipStream_t st);
int launch_jac_coord_f32(ba_problem *p, const float *d_x, float *d_vals, hipStream_t st);

// ---- launchers (ba_normal_kernels.hip) ----------------------------------------------------------
// point side: H_pp (6/pt: xx,xy,xz,yy,yz,zz) and g_p (3/pt) from J, r.  Either output may be null.
int launch_point_blocks(ba_problem *p, const double *d_J, const double *d_r, double *d_Hpp, double *d_gp,
                        hipStream_t st);
// camera side: H_cc (45/cam packed lower row-major) and g_c (9/cam).  Either output may be null.
int launch_cam_blocks(ba_problem *p, const double *d_J, const double *d_r, double *d_Hcc, double *d_gc,
                      hipStream_t st);

// ---- dense LDL^T (ba_dense_ldl.hip) ---------------------------------------------------------------
template <typename T>
int dense_ldl_alloc(DenseLDLT<T> *w, int64_t n_unpadded, T *external_S, int world = 1, int rank = 0, bool lazy_S = false,
                    bool own_only = false);
// lazy_S: the tiles of S (and the panel buffers V) are left out until dense_ldl_alloc_S
template <typename T>
int dense_ldl_alloc_S(DenseLDLT<T> *w);
template <typename T>
void dense_ldl_free(DenseLDLT<T> *w);
int64_t dense_ldl_tiles_doubles(int64_t n_unpadded);  // number of ELEMENTS of the packed lower tiles
// factor S in place (L below the diagonal tiles' diagonal, D separately); *zero_pivot set on exact zero pivot.
// d_b != null: the forward substitution L y = b of that right-hand side (length nt*NB, clobbered) is fused into the
// panel solves; pass forward_done = true to dense_ldl_solve afterwards.
template <typename T>
int dense_ldl_factor(ba_problem *p, DenseLDLT<T> *w, hipStream_t st, int *zero_pivot, T *d_b);
// the same factorisation with the tile column pairs distributed over the ranks of p->comm (owner of pair q: q % world):
// on entry rank r holds the (summed) tile columns it owns, on exit every rank holds the complete factor (L, Linv, D).
// No fused forward substitution: call dense_ldl_solve(..., forward_done = false).
template <typename T>
int dense_ldl_factor_dist(ba_problem *p, DenseLDLT<T> *w, hipStream_t st, T *d_b = nullptr);
// block-sparse S: the use of a tile pattern (tile_pattern_build, ba_order.h) by a workspace (null: dense).  The pattern object
// must outlive the workspace.
template <typename T>
int dense_ldl_use_pattern(DenseLDLT<T> *w, const TilePattern *pat);
// solve S x = b for one right-hand side held in d_b (length nt*NB, overwritten by x)
template <typename T>
int dense_ldl_solve(ba_problem *p, DenseLDLT<T> *w, T *d_b, hipStream_t st, bool forward_done);

// ---- transport (ba_comm.hip): all on stream st, in place, no-ops without a communicator --------------------
int comm_allreduce(ba_problem *p, double *d_buf, int64_t count, hipStream_t st);
int comm_reduce(ba_problem *p, double *d_buf, int64_t count, int root, hipStream_t st);  // sum lands on root only
int comm_reduce_f32(ba_problem *p, float *d_buf, int64_t count, int root, hipStream_t st);
int comm_bcast(ba_problem *p, void *d_buf, int64_t bytes, int root, hipStream_t st);
// d_buf: world segments of `count` elements (Float64, or Float32 when f32); segment `rank` receives the sum, in place
int comm_reduce_scatter(ba_problem *p, void *d_buf, int64_t count, bool f32, hipStream_t st);
int comm_group_begin(ba_problem *p);  // RCCL: fuse the calls up to comm_group_end into one launch
int comm_group_end(ba_problem *p);
void comm_free(ba_problem *p);

// ---- LM (ba_lm.hip) ---------------------------------------------------------------------------------
void lm_free(ba_problem *p);
