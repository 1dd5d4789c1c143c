// Levenberg-Marquardt controller (host) over the device kernels.
//
// Control flow, constants and stopping tests follow the reference line by line:
//   variant 1: src/lm.jl:15-418            (lambda0 = max(lambda, 1e10/|J'r|), ared >= 1e-4 pred, line search)
//   variant 0: src/LevenbergMarquardt.jl:16-385 (what src/solve_ba.jl runs)
// What differs is how the linear step is obtained: the reference factors the augmented matrix
// K = [[I J];[J' -lambda I]] (src/lm.jl:68-100,154-238); here the residual rows and point columns of K are
// eliminated in closed form on the device (ba_normal_kernels.hip) and the remaining reduced camera system is
// factored densely on the f64 matrix cores (ba_dense_ldl.hip).  Both give (J'J + lambda I) delta = -J' r and
// 1/2|delta_r|^2 = 1/2|J delta + r|^2 (K [dr; d] = [-r; 0]  <=>  dr = -(r + J d)), which is evaluated directly.
//
// One device->host copy of a handful of scalars per iteration drives the accept/reject logic, as the reference's
// norm() calls do.  As in the reference, a rejected step only changes the damping: J, Hpp, Hcc, gp, gc are reused.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "ba_internal.h"
#include "ba_lm_internal.h"

namespace {

double wall() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// scalar slots.  "sharded" sums are partial per rank and all-reduced; "replicated" ones are identical on every rank.
// slots 0..2 are refreshed (and reduced) with the linearisation, slots 3..5 with every trial step
enum { SH_RSQ = 0, SH_GP, SH_X_P, SH_RSQ_TRIAL, SH_MODEL, SH_DELTA_P, SH_COUNT = 8 };
constexpr int SH_LIN_COUNT = 3, SH_TRIAL_FIRST = 3, SH_TRIAL_COUNT = 3;
enum { RP_DELTA_C = 0, RP_X_C, RP_GC, RP_COUNT = 8 };

template <typename T>
int dmalloc(T **ptr, int64_t count) {
  BA_HIP_CHECK(hipMalloc((void **)ptr, (size_t)(count > 0 ? count : 1) * sizeof(T)));
  return BA_OK;
}

template <typename T>
int upload_vec(T **d, const std::vector<T> &h) {
  BA_CHECK(dmalloc(d, (int64_t)h.size()));
  if (!h.empty()) BA_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return BA_OK;
}

// Build the (camera_a >= camera_b)-sorted list of observation pairs sharing a point.  "Camera" here is the BLOCK ROW of S the
// camera sits at: pos[c] under a fill-reducing camera ordering (empty: c itself).
int build_tasks(ba_problem *p, SchurTasks *T, const std::vector<int> &pos) {
  const int64_t ncams = p->ncams, npnts = p->npnts;
  const std::vector<int> &ptr = p->h_pt_ptr, &obs = p->h_pt_obs;
  std::vector<int> cam_at;
  if (!pos.empty()) {
    cam_at.resize(p->h_cam0.size());
    for (size_t o = 0; o < cam_at.size(); o++) cam_at[o] = pos[(size_t)p->h_cam0[o]];
  }
  const std::vector<int> &cam = pos.empty() ? p->h_cam0 : cam_at;
  // pass 1: tasks per camera_a
  std::vector<int64_t> ca_ptr((size_t)ncams + 1, 0);
  int64_t ntasks = 0;
  for (int64_t pt = 0; pt < npnts; pt++)
    for (int qa = ptr[(size_t)pt]; qa < ptr[(size_t)pt + 1]; qa++) {
      int ca = cam[(size_t)obs[(size_t)qa]];
      for (int qb = ptr[(size_t)pt]; qb < ptr[(size_t)pt + 1]; qb++)
        if (ca >= cam[(size_t)obs[(size_t)qb]]) {
          ca_ptr[(size_t)ca + 1]++;
          ntasks++;
        }
    }
  if (ntasks > (int64_t)2000000000) {
    ba_set_error("Schur task list too long (%lld)", (long long)ntasks);
    return BA_ERR_ARG;
  }
  for (int64_t c = 0; c < ncams; c++) ca_ptr[(size_t)c + 1] += ca_ptr[(size_t)c];
  // pass 2: bucket by camera_a (point order preserved)
  std::vector<int> ta((size_t)ntasks), tb((size_t)ntasks);
  {
    std::vector<int64_t> cur(ca_ptr.begin(), ca_ptr.end() - 1);
    for (int64_t pt = 0; pt < npnts; pt++)
      for (int qa = ptr[(size_t)pt]; qa < ptr[(size_t)pt + 1]; qa++) {
        int oa = obs[(size_t)qa], ca = cam[(size_t)oa];
        for (int qb = ptr[(size_t)pt]; qb < ptr[(size_t)pt + 1]; qb++) {
          int ob = obs[(size_t)qb];
          if (ca >= cam[(size_t)ob]) {
            int64_t q = cur[(size_t)ca]++;
            ta[(size_t)q] = oa;
            tb[(size_t)q] = ob;
          }
        }
      }
  }
  // pass 3: inside each camera_a bucket, stable counting sort by camera_b; emit keys (diagonal key always)
  std::vector<int> sa((size_t)ntasks), sb((size_t)ntasks), key_ptr, key_ca, key_cb;
  std::vector<int> cnt((size_t)ncams + 1);
  key_ptr.push_back(0);
  for (int64_t ca = 0; ca < ncams; ca++) {
    const int64_t b0 = ca_ptr[(size_t)ca], b1 = ca_ptr[(size_t)ca + 1];
    std::fill(cnt.begin(), cnt.begin() + ca + 2, 0);
    for (int64_t q = b0; q < b1; q++) cnt[(size_t)cam[(size_t)tb[(size_t)q]] + 1]++;
    for (int64_t cb = 0; cb <= ca; cb++) {
      int c = cnt[(size_t)cb + 1];
      if (c > 0 || cb == ca) {
        key_ca.push_back((int)ca);
        key_cb.push_back((int)cb);
        key_ptr.push_back(key_ptr.back() + c);
      }
      cnt[(size_t)cb + 1] += cnt[(size_t)cb];
    }
    // cnt[cb] = offset of camera_b bucket inside [b0, b1)
    for (int64_t q = b0; q < b1; q++) {
      int cb = cam[(size_t)tb[(size_t)q]];
      int64_t dst = b0 + cnt[(size_t)cb]++;
      sa[(size_t)dst] = ta[(size_t)q];
      sb[(size_t)dst] = tb[(size_t)q];
    }
  }
  T->nkeys = (int64_t)key_ca.size();
  T->ntasks = ntasks;
  {  // which 128 x 128 tiles of S receive a 9 x 9 block (a block straddles at most two tile rows and two tile columns)
    const int64_t nt = std::max<int64_t>(1, (9 * ncams + NB - 1) / NB);
    T->tile_occ.assign((size_t)(nt * nt), 0);
    for (size_t q = 0; q < key_ca.size(); q++) {
      const int64_t r0 = 9 * (int64_t)key_ca[q], c0 = 9 * (int64_t)key_cb[q];
      for (int64_t ti = r0 / NB; ti <= (r0 + 8) / NB; ti++)
        for (int64_t tj = c0 / NB; tj <= (c0 + 8) / NB; tj++)
          if (ti >= tj) T->tile_occ[(size_t)(ti * nt + tj)] = 1;
    }
  }
  {  // chunking of the long keys (see SchurTasks)
    // chunk size: the partial blocks cost traffic, so as large as leaves ~8 k chunks for the chip (sweeps on MI355X: LadyBug-49
    // 4 / 8 / 16 / 32 -> 0.107 / 0.070 / 0.058 / 0.065 ms; Dubrovnik-356 8 / 32 / 128 / 256 -> 1.12 / 0.68 / 0.58 / 0.57 ms
    // (unsplit: 2.26); Venice-1778 unsplit / 8 / 32 / 128 -> 4.36 / 4.02 / 3.52 / 3.42 ms)
    int ch = 16;
    while (ch < 128 && ntasks / (2 * ch) >= 8192) ch *= 2;
    if (const char *e = getenv("BA_SCHUR_CHUNK")) ch = atoi(e);  // 0 disables the split
    T->chunk = ch;
    std::vector<int> skey, skey_c0, ct0, ct1;
    if (ch > 0)
      for (int64_t k = 0; k < T->nkeys; k++) {
        const int t0 = key_ptr[(size_t)k], t1 = key_ptr[(size_t)k + 1];
        if (t1 - t0 <= 2 * ch) continue;
        skey.push_back((int)k);
        skey_c0.push_back((int)ct0.size());
        for (int t = t0; t < t1; t += ch) {
          ct0.push_back(t);
          ct1.push_back(t + ch < t1 ? t + ch : t1);
        }
      }
    skey_c0.push_back((int)ct0.size());
    T->nsplit = (int64_t)skey.size();
    T->nchunks = (int64_t)ct0.size();
    T->h_skey = skey;
    if (T->nsplit > 0) {
      BA_CHECK(upload_vec(&T->skey, skey));
      BA_CHECK(upload_vec(&T->skey_c0, skey_c0));
      BA_CHECK(upload_vec(&T->chunk_t0, ct0));
      BA_CHECK(upload_vec(&T->chunk_t1, ct1));
      BA_CHECK(dmalloc(&T->partial, 81 * T->nchunks));
    }
  }
  T->h_key_cb = key_cb;
  BA_CHECK(upload_vec(&T->key_ptr, key_ptr));
  BA_CHECK(upload_vec(&T->key_ca, key_ca));
  BA_CHECK(upload_vec(&T->key_cb, key_cb));
  BA_CHECK(upload_vec(&T->task_a, sa));
  BA_CHECK(upload_vec(&T->task_b, sb));
  return BA_OK;
}

// one device buffer [rhs(npad) | gc(npad) | hdiag(npad) | SH_COUNT scalars]: gc, hdiag and the first scalars are one
// all-reduce.  (The tiles of S are their own allocation, made when the first direct solve needs them: ensure_dense.)
int64_t reduce_layout(ba_problem *p, int64_t *off_rhs, int64_t *off_gc, int64_t *off_scal) {
  const int64_t n = 9 * p->ncams;
  const int64_t npad = ((n + NB - 1) / NB > 0 ? (n + NB - 1) / NB : 1) * NB;
  if (off_rhs) *off_rhs = 0;
  if (off_gc) *off_gc = npad;
  if (off_scal) *off_scal = 3 * npad;
  return 3 * npad + SH_COUNT;
}

struct LMState {
  double *red = nullptr;  // [rhs | gc | hdiag | sharded scalars]
  bool own_red = false;
  int64_t off_rhs = 0, off_gc = 0, off_scal = 0, red_doubles = 0;
  double *scal_rep = nullptr;
  double *h_sh = nullptr, *h_rp = nullptr;  // pinned
};

}  // namespace

struct LMWorkFull : LMWork {
  LMState s;
  std::vector<SchurChunk> chunks;  // per-rank ownership of S: chunks of tile columns, assembled and reduced one by one
  double *stage = nullptr;         // the chunk being assembled for another owner (stage_tiles tiles)
  float *stage32 = nullptr;        // its Float32 copy when the reduce travels in Float32
  int64_t stage_tiles = 0;         // (reduce-scatter assembly: all the staging tiles, stage_bufs buffers of stage_tiles / stage_bufs)
  bool assembly_rs = false;        // chunks are reduce-scattered (one segment per owner) instead of reduced onto one owner
  int stage_bufs = 1;              // 2: chunk c+1 is assembled while chunk c travels (transfer stream)
  hipEvent_t ev_stage_ready[2] = {nullptr, nullptr}, ev_stage_free[2] = {nullptr, nullptr};
  TilePattern pattern;       // tile pattern of S after the symbolic factorisation (ensure_dense)
  bool use_pattern = false;  // the block-sparse list schedule is in use on this handle
  // fill-reducing camera ordering of the reduced camera system (`perm` of the reference's solvers, src/lm.jl:84-88): the
  // method asked for, the camera sequence it gave (block row k of S holds camera h_cam_perm[k]; empty: the caller's
  // numbering), the name of the candidate that won
  int order_method = BA_ORDER_AMD;
  std::vector<int> h_cam_perm;
  const char *order_name = "natural";
  int order_split = 0;  // the sequence eliminates from both ends: tile column pair at which the second run starts (0: one run)
  // facto_type = Float32 (src/lm.jl:170-173): Float32 copy of the reduced camera system, allocated on first use
  DenseLDLT<float> ldl32;
  float *rhs32 = nullptr;
  bool have32 = false, last_f32 = false;
  // facto_type = Float16 (src/lm.jl:165-169): set per solve; J_lin / r_lin / cr0: what the model value of the current step is
  // evaluated on (the Float16-rounded scaled copies in that mode, J and r otherwise), see linear_step
  bool f16 = false;
  // what the model value of a step is evaluated on: the Float16-rounded scaled copies in that branch, J and r otherwise
  // (looked up at call time: w->r / w->r_trial swap on accepted steps, a recorded graph must not pin them)
  const double *J_lin() const { return f16 ? Jq : J; }
  const double *r_lin() const { return f16 ? rq : r; }
  double cr0() const { return f16 ? 1.0 / (0.1 * 6.55e4) : 1.0; }
  // eltype(x) = Float32 runs (BALNLPModel(file, Float32), src/BALNLPModels.jl:91): x, r and J are produced by the Float32
  // kernels and widened; every iterate is rounded to Float32.  Buffers allocated on first use.
  float *xf = nullptr, *rf = nullptr, *Jf = nullptr;
  // hipGraph replay of the two launch sequences of the LM loop (launch-bound on small problems: LadyBug-49 issues ~60
  // kernels of a few microseconds per iteration).  x/x_trial and r/r_trial swap on an accepted step, so each sequence
  // is recorded once per parity of the swap; the damping reaches the recorded kernels through d_lambda.
  double *d_lambda = nullptr, *h_lambda = nullptr;  // device scalar, pinned staging
  int *h_flag = nullptr;                            // pinned copy of the pivot flag
  hipGraphExec_t g_step[2] = {nullptr, nullptr}, g_refresh[2] = {nullptr, nullptr};
  int g_key = -1;  // normalize + 4 * facto_f32 + 8 * x_f32 the graphs were recorded for
  int parity = 0;
  bool g_off = false;  // a recording failed on this handle: plain launches from then on
  // facto = PCG: block-Jacobi preconditioned conjugate gradients on the reduced camera system, S never formed (pcg_solve).
  // Buffers allocated on first use: iterate, residual, preconditioned residual, direction, S * direction, W U^-1 W' part
  // (n each), point-side intermediate and a zero vector (3 npnts), the 9 x 9 diagonal blocks (45 per camera).
  bool pcg = false;
  double pcg_tol = 1e-8;
  int pcg_maxit = 0;  // 0: default
  int64_t n_cg = 0;   // CG iterations of the current solve
  double *cgx = nullptr, *cgr = nullptr, *cgz = nullptr, *cgp = nullptr, *cgq = nullptr, *cgt = nullptr;
  double *cgh = nullptr, *zero3 = nullptr, *blk45 = nullptr, *cg_scal = nullptr, *h_cg = nullptr;
};

namespace {
template <typename A, typename B>
__global__ void k_convert(const A *__restrict__ in, B *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (B)in[i];
}
template <typename A, typename B>
int launch_convert(const A *in, B *out, int64_t n, hipStream_t st) {
  if (n <= 0) return BA_OK;
  const int64_t blocks = std::min<int64_t>((n + 255) / 256, 1 << 16);
  hipLaunchKernelGGL((k_convert<A, B>), dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
  BA_HIP_CHECK(hipGetLastError());
  return BA_OK;
}
}  // namespace

// BA_DIST_FACTOR=0: keep the whole reduced camera system on every rank (one all-reduce of S, replicated factorisation)
static bool dist_factor_on(ba_problem *p) {
  static const bool off = [] { const char *e = getenv("BA_DIST_FACTOR"); return e && e[0] == '0'; }();
  return p->comm.active() && !off;
}

static int ensure_xf32(ba_problem *p, LMWorkFull *w) {
  if (w->xf) return BA_OK;
  BA_HIP_CHECK(hipMalloc((void **)&w->xf, (size_t)(w->nvar > 0 ? w->nvar : 1) * sizeof(float)));
  BA_HIP_CHECK(hipMalloc((void **)&w->rf, (size_t)(w->nequ > 0 ? w->nequ : 1) * sizeof(float)));
  BA_HIP_CHECK(hipMalloc((void **)&w->Jf, (size_t)(p->nobs > 0 ? 24 * p->nobs : 1) * sizeof(float)));
  return BA_OK;
}

static int ensure_f32(LMWorkFull *w) {
  if (w->have32) return BA_OK;
  BA_CHECK(dense_ldl_alloc<float>(&w->ldl32, w->n, nullptr, w->ldl.world, w->ldl.rank, false, w->ldl.own_only));
  if (w->ldl.own_only && !w->stage32) BA_HIP_CHECK(hipMalloc((void **)&w->stage32, (size_t)std::max<int64_t>(1, w->stage_tiles) * NB * NB * sizeof(float)));
  if (w->use_pattern) BA_CHECK(dense_ldl_use_pattern(&w->ldl32, &w->pattern));
  BA_HIP_CHECK(hipMalloc((void **)&w->rhs32, (size_t)w->npad * sizeof(float)));
  w->have32 = true;
  return BA_OK;
}

static int ensure_f16(ba_problem *p, LMWorkFull *w) {
  if (w->Jq) return BA_OK;
  BA_CHECK(dmalloc(&w->jn2, w->nvar));
  BA_CHECK(dmalloc(&w->dcol, w->nvar));
  BA_CHECK(dmalloc(&w->damp, w->nvar));
  BA_CHECK(dmalloc(&w->Jq, 24 * p->nobs));
  BA_CHECK(dmalloc(&w->rq, w->nequ));
  return BA_OK;
}

static int lm_ensure(ba_problem *p) {
  if (p->lm) return BA_OK;
  LMWorkFull *w = new LMWorkFull();
  p->lm = w;
  const int64_t ncams = p->ncams, npnts = p->npnts, nobs = p->nobs;
  w->nvar = 9 * ncams + 3 * npnts;
  w->nequ = 2 * nobs;
  w->n = 9 * ncams;
  w->s.red_doubles = reduce_layout(p, &w->s.off_rhs, &w->s.off_gc, &w->s.off_scal);
  BA_CHECK(dmalloc(&w->s.red, w->s.red_doubles));
  w->s.own_red = true;
  // with a communicator the tile column pairs of S are laid out by owner rank (one contiguous range per rank).  The tiles
  // themselves (n^2/2 doubles: 1 GB for Venice, 60 GB for Final-13682), the Schur task list and the per-observation Y blocks
  // are allocated by ensure_dense when a direct solve first needs them: a handle that only ever runs facto = :PCG never
  // holds anything of the size of S.
  // ... and with the distributed factorisation a rank holds ONLY its own tile columns of S (per-rank ownership): the other
  // ranks' contributions pass through a staging buffer of at most half that size, chunk by chunk (ensure_dense, linear_step)
  BA_CHECK(dense_ldl_alloc(&w->ldl, w->n, (double *)nullptr, p->comm.active() ? p->comm.world : 1, p->comm.active() ? p->comm.rank : 0, true,
                           dist_factor_on(p)));
  w->npad = w->ldl.n;
  w->rhs = w->s.red + w->s.off_rhs;
  w->gc = w->s.red + w->s.off_gc;
  w->hdiag = w->gc + w->npad;
  w->scal = w->s.red + w->s.off_scal;
  BA_HIP_CHECK(hipMemset(w->s.red + w->s.off_rhs, 0, (size_t)(w->s.red_doubles - w->s.off_rhs) * sizeof(double)));
  BA_HIP_CHECK(hipDeviceSynchronize());  // (null-stream memset: not ordered against the handle's non-blocking stream)
  BA_CHECK(dmalloc(&w->x, w->nvar));
  BA_CHECK(dmalloc(&w->x_trial, w->nvar));
  BA_CHECK(dmalloc(&w->delta, w->nvar));
  BA_CHECK(dmalloc(&w->r, w->nequ));
  BA_CHECK(dmalloc(&w->r_trial, w->nequ));
  BA_CHECK(dmalloc(&w->J, 24 * nobs));
  BA_CHECK(dmalloc(&w->Hpp, 6 * npnts));
  BA_CHECK(dmalloc(&w->gp, 3 * npnts));
  BA_CHECK(dmalloc(&w->Uinv, 6 * npnts));
  BA_CHECK(dmalloc(&w->u, 3 * npnts));
  BA_CHECK(dmalloc(&w->Hcc, 45 * ncams));
  BA_CHECK(dmalloc(&w->colscale, 9 * ncams));
  BA_CHECK(dmalloc(&w->partial, std::max<int64_t>(RED_BLOCKS, (npnts + 255) / 256)));  // k_wtv<true>: one partial per 256 points
  BA_CHECK(dmalloc(&w->partial_multi, (int64_t)SUMSQ_JOBS * RED_BLOCKS));
  BA_HIP_CHECK(hipMalloc((void **)&w->cam_pnt, (size_t)(p->nobs > 0 ? p->nobs : 1) * sizeof(int)));
  BA_CHECK(launch_cam_pnt(p, w->cam_pnt, p->stream));
  BA_CHECK(dmalloc(&w->s.scal_rep, (int64_t)RP_COUNT));
  BA_HIP_CHECK(hipHostMalloc((void **)&w->s.h_sh, SH_COUNT * sizeof(double)));
  BA_HIP_CHECK(hipHostMalloc((void **)&w->s.h_rp, RP_COUNT * sizeof(double)));
  BA_HIP_CHECK(hipHostMalloc((void **)&w->h_lambda, sizeof(double)));
  BA_HIP_CHECK(hipHostMalloc((void **)&w->h_flag, sizeof(int)));
  BA_CHECK(dmalloc(&w->d_lambda, (int64_t)1));
  return BA_OK;
}

// Per-rank ownership of S: the tile columns of every rank (one contiguous range of the owner-major layout) are cut into
// chunks of at most about half a rank's share; a chunk is assembled -- by every rank, from its own observations -- either
// straight into the owner's tiles (the owner itself) or into the staging buffer, and reduced onto the owner.  So no rank
// ever holds more of S than its own columns plus one chunk: <= 1.5 |S| / world (plus the panel buffers of the
// factorisation).  A key's 9 x 9 block can straddle two tile columns: it is listed with both chunks and each stores the
// elements that fall into its own columns (the offset table of the chunk marks the others).
static int upload_chunk_tables(LMWorkFull *w, std::vector<std::vector<int64_t>> &cco, const std::vector<int> &col_chunk_of_col,
                               bool sparse);

// Reduce-scatter form of the chunked assembly (default; BA_ASSEMBLY=reduce keeps the chunk-onto-one-owner form).  A chunk
// reduced onto ONE owner is a many-to-one transfer: over point-to-point xGMI it moves at one link's rate.  Here a chunk is
// one slice of EVERY owner's tile columns -- `world` segments of `seg` tiles in the staging buffer, segment r a slice of rank
// r's storage (whole tile columns; shorter slices zero-padded) -- and ONE in-place reduce-scatter delivers to every owner
// its segment: all links of every GPU carry 1 / world of the chunk.  The owner then copies its segment into its tiles.  With
// two staging buffers chunk c travels on the transfer stream while chunk c+1 is assembled on the main one.  Slices per
// owner: as many as keep all staging within about half a rank's share of S (4 world with two buffers), never finer than a
// tile column; when two buffers would not fit (small problems) there is one and the transfer is in line.
static int build_chunks_rs(ba_problem *p, LMWorkFull *w) {
  const DenseLDL &l = w->ldl;
  const int64_t nt = l.nt;
  const int P = l.world, me = l.rank;
  const bool sparse = w->use_pattern;
  auto col_tiles = [&](int64_t j) { return sparse ? l.h_col_cnt[(size_t)j] : nt - j; };
  std::vector<int64_t> share((size_t)P, 0), local_off((size_t)nt, 0);
  int64_t max_col = 1;
  for (int r = 0; r < P; r++) {
    int64_t run = 0;
    for (int64_t j = 0; j < nt; j++)
      if ((j / 2) % P == r) {
        local_off[(size_t)j] = run;
        run += col_tiles(j);
        max_col = std::max(max_col, col_tiles(j));
      }
    share[(size_t)r] = run;
  }
  const int64_t max_share = *std::max_element(share.begin(), share.end());
  const int64_t budget = std::max<int64_t>(max_share / 2, 1);  // all staging together
  auto plan = [&](int nsl, std::vector<std::vector<std::pair<int64_t, int64_t>>> *slices, int64_t *seg_out) {
    // per owner: its columns in order cut into nsl runs of about share / nsl tiles; slices[r][c] = (first column index in
    // the owner's list, one past the last); seg = the longest run
    slices->assign((size_t)P, {});
    int64_t seg = 1;
    for (int r = 0; r < P; r++) {
      std::vector<int64_t> cols;
      for (int64_t j = 0; j < nt; j++)
        if ((j / 2) % P == r) cols.push_back(j);
      // cumulative boundaries: slice c ends with the first column at which the running count reaches (c + 1) / nsl of the
      // share, so no slice exceeds its even part by more than one column
      size_t a = 0;
      int64_t run = 0;
      for (int c = 0; c < nsl; c++) {
        const int64_t goal = (share[(size_t)r] * (c + 1) + nsl - 1) / nsl;
        size_t b = a;
        int64_t n = 0;
        while (b < cols.size() && (run + n < goal || c == nsl - 1)) n += col_tiles(cols[b++]);
        (*slices)[(size_t)r].push_back({(int64_t)a, (int64_t)b});
        seg = std::max(seg, n);
        run += n;
        a = b;
      }
    }
    *seg_out = seg;
  };
  std::vector<std::vector<std::pair<int64_t, int64_t>>> slices;
  int64_t seg = 0;
  int nsl = 4 * P, bufs = 2;
  plan(nsl, &slices, &seg);
  if (2 * P * seg > budget + 2 * nt) bufs = 1;  // two buffers do not fit (slices are whole tile columns): one, the transfer in line
  w->assembly_rs = true;
  w->stage_bufs = bufs;
  w->stage_tiles = (int64_t)bufs * P * seg;
  w->chunks.clear();
  std::vector<std::vector<int64_t>> cco;
  std::vector<int> col_chunk((size_t)nt, -1);
  for (int c = 0; c < nsl; c++) {
    SchurChunk ch;
    ch.owner = -1;
    ch.seg = seg;
    ch.ntiles = (int64_t)P * seg;
    std::vector<int64_t> table((size_t)nt, BA_NO_TILE);
    bool any = false;
    for (int r = 0; r < P; r++) {
      std::vector<int64_t> cols;
      for (int64_t j = 0; j < nt; j++)
        if ((j / 2) % P == r) cols.push_back(j);
      const auto sl = slices[(size_t)r][(size_t)c];
      if (sl.first >= sl.second) continue;
      any = true;
      const int64_t t0 = local_off[(size_t)cols[(size_t)sl.first]];
      int64_t n = 0;
      for (int64_t a = sl.first; a < sl.second; a++) {
        const int64_t j = cols[(size_t)a];
        table[(size_t)j] = (int64_t)r * seg + (local_off[(size_t)j] - t0);
        col_chunk[(size_t)j] = (int)w->chunks.size();
        n += col_tiles(j);
      }
      if (r == me) {
        ch.my_t0 = t0;
        ch.my_n = n;
      }
    }
    if (!any) continue;
    w->chunks.push_back(ch);
    cco.push_back(table);
  }
  for (int q = 0; q < 2; q++) {
    if (!w->ev_stage_ready[q]) BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_stage_ready[q], hipEventDisableTiming));
    if (!w->ev_stage_free[q]) BA_HIP_CHECK(hipEventCreateWithFlags(&w->ev_stage_free[q], hipEventDisableTiming));
  }
  return upload_chunk_tables(w, cco, col_chunk, sparse);
}

static int build_chunks(ba_problem *p, LMWorkFull *w) {
  {
    const char *e = getenv("BA_ASSEMBLY");  // reduce: every chunk onto ONE owner (round 3's form); default: reduce-scatter
    if (!(e && e[0] == 'r' && e[1] == 'e' && e[2] == 'd' && e[3] == 'u' && e[4] == 'c' && e[5] == 'e' && e[6] == 0)) return build_chunks_rs(p, w);
  }
  const DenseLDL &l = w->ldl;
  const int64_t nt = l.nt;
  const int P = l.world;
  const SchurTasks &T = w->tasks;
  // block-sparse S: a tile column holds the pattern's tiles only (h_col_cnt) and the chunk tables are compressed like the
  // workspace's own (head nt, column offsets, the shared nt x nt row positions)
  const bool sparse = w->use_pattern;
  auto col_tiles = [&](int64_t j) { return sparse ? l.h_col_cnt[(size_t)j] : nt - j; };
  std::vector<int> col_chunk((size_t)nt, -1);
  std::vector<std::vector<int64_t>> cco;
  w->chunks.clear();
  w->stage_tiles = 0;
  for (int r = 0; r < P; r++) {
    const int64_t t_begin = l.own_range[(size_t)r], t_end = l.own_range[(size_t)r + 1], share = t_end - t_begin;
    const int64_t target = std::max<int64_t>(1, (share + 1) / 2);
    SchurChunk c;
    c.owner = r;
    c.t0 = 0;
    std::vector<int64_t> table((size_t)nt, BA_NO_TILE);
    int64_t local = 0;  // tile offset inside rank r's own layout
    auto flush = [&]() {
      if (c.ntiles == 0) return;
      w->chunks.push_back(c);
      cco.push_back(table);
      if (r != l.rank) w->stage_tiles = std::max(w->stage_tiles, c.ntiles);
      c = SchurChunk();
      c.owner = r;
      c.t0 = local;
      std::fill(table.begin(), table.end(), BA_NO_TILE);
    };
    for (int64_t j = 0; j < nt; j++) {
      if ((j / 2) % P != r) continue;
      const int64_t colt = col_tiles(j);
      if (c.ntiles > 0 && c.ntiles + colt > target) flush();
      table[(size_t)j] = local - c.t0;  // offset of tile (j, j) inside the chunk's destination buffer
      col_chunk[(size_t)j] = (int)w->chunks.size();
      c.ntiles += colt;
      local += colt;
    }
    flush();
  }
  return upload_chunk_tables(w, cco, col_chunk, sparse);
}

// the keys of every chunk, its offset table on the device, the staging buffer
static int upload_chunk_tables(LMWorkFull *w, std::vector<std::vector<int64_t>> &cco, const std::vector<int> &col_chunk, bool sparse) {
  const DenseLDL &l = w->ldl;
  const int64_t nt = l.nt;
  const SchurTasks &T = w->tasks;
  // keys per chunk (a block touches tile columns 9 cb / NB and (9 cb + 8) / NB)
  const size_t nc = w->chunks.size();
  std::vector<std::vector<int>> keys(nc), skeys(nc);
  std::vector<int> split_index(T.h_key_cb.size(), -1);
  for (size_t q = 0; q < T.h_skey.size(); q++) split_index[(size_t)T.h_skey[q]] = (int)q;
  for (size_t k = 0; k < T.h_key_cb.size(); k++) {
    const int64_t c0 = 9 * (int64_t)T.h_key_cb[k];
    const int ch0 = col_chunk[(size_t)(c0 / NB)], ch1 = col_chunk[(size_t)((c0 + 8) / NB)];
    for (int ch : {ch0, ch1 == ch0 ? -1 : ch1}) {
      if (ch < 0) continue;
      if (split_index[k] >= 0) skeys[(size_t)ch].push_back(split_index[k]);
      else keys[(size_t)ch].push_back((int)k);
    }
  }
  for (size_t c = 0; c < nc; c++) {
    SchurChunk &ch = w->chunks[c];
    {  // (tix reads the head of a table one entry before its pointer: 0 = dense columns, nt = compressed ones)
      std::vector<int64_t> with_head(cco[c].size() + 1 + (sparse ? (size_t)(nt * nt) : 0), 0);
      std::copy(cco[c].begin(), cco[c].end(), with_head.begin() + 1);
      if (sparse) {
        with_head[0] = nt;
        std::copy(l.h_col_tab.begin() + 1 + nt, l.h_col_tab.end(), with_head.begin() + 1 + nt);
      }
      BA_CHECK(upload_vec(&ch.cco_alloc, with_head));
      ch.cco = ch.cco_alloc + 1;
    }
    BA_CHECK(upload_vec(&ch.keys, keys[c]));
    BA_CHECK(upload_vec(&ch.skeys, skeys[c]));
    ch.nkeys = (int64_t)keys[c].size();
    ch.nskeys = (int64_t)skeys[c].size();
  }
  BA_HIP_CHECK(hipMalloc((void **)&w->stage, (size_t)std::max<int64_t>(1, w->stage_tiles) * NB * NB * sizeof(double)));
  return BA_OK;
}

// OR of a flag array over the ranks (set-up only): as doubles through the all-reduce the transport has
static int allreduce_flags(ba_problem *p, std::vector<unsigned char> *flags) {
  if (!p->comm.active() || flags->empty()) return BA_OK;
  std::vector<double> h(flags->begin(), flags->end());
  double *d = nullptr;
  BA_HIP_CHECK(hipMalloc((void **)&d, h.size() * sizeof(double)));
  BA_HIP_CHECK(hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  int rc = comm_allreduce(p, d, (int64_t)h.size(), p->stream);
  if (rc == BA_OK) {
    hipError_t e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) {
      ba_set_error("allreduce_flags: %s", hipGetErrorString(e));
      rc = BA_ERR_HIP;
    }
  }
  (void)hipFree(d);
  BA_CHECK(rc);
  for (size_t i = 0; i < h.size(); i++) (*flags)[i] = h[i] != 0.0;
  return BA_OK;
}

// OR of the camera graph over the ranks: a rank's observations only show the camera pairs ITS points connect, the ordering
// must be that of the whole problem and the same on every rank.  The bit rows travel as counts, eight 6-bit fields per
// double (a field holds at most `world` <= 63), in slices of at most 128 MB.
static int allreduce_cam_graph(ba_problem *p, CamGraph *g) {
  if (!p->comm.active() || g->bits.empty()) return BA_OK;
  const size_t nwords = g->bits.size(), ndbl = nwords * 8;
  const size_t slice = std::min<size_t>(ndbl, (size_t)16 << 20);
  std::vector<double> h(slice);
  double *d = nullptr;
  BA_HIP_CHECK(hipMalloc((void **)&d, slice * sizeof(double)));
  int rc = BA_OK;
  for (size_t off = 0; off < ndbl && rc == BA_OK; off += slice) {
    const size_t cnt = std::min(slice, ndbl - off);
    for (size_t i = 0; i < cnt; i++) {  // double off + i: byte ((off + i) & 7) of word (off + i) / 8
      const uint64_t byte = (g->bits[(off + i) >> 3] >> (8 * ((off + i) & 7))) & 0xff;
      uint64_t packed = 0;
      for (int b = 0; b < 8; b++) packed |= ((byte >> b) & 1) << (6 * b);
      h[i] = (double)packed;
    }
    hipError_t e = hipMemcpyAsync(d, h.data(), cnt * sizeof(double), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) rc = comm_allreduce(p, d, (int64_t)cnt, p->stream);
    if (e == hipSuccess && rc == BA_OK) e = hipMemcpyAsync(h.data(), d, cnt * sizeof(double), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && rc == BA_OK) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) {
      ba_set_error("allreduce_cam_graph: %s", hipGetErrorString(e));
      rc = BA_ERR_HIP;
    }
    if (rc != BA_OK) break;
    for (size_t i = 0; i < cnt; i++) {
      const uint64_t packed = (uint64_t)h[i];
      uint64_t byte = 0;
      for (int b = 0; b < 8; b++) byte |= (uint64_t)(((packed >> (6 * b)) & 63) != 0) << b;
      uint64_t &word = g->bits[(off + i) >> 3];
      const int sh = 8 * (int)((off + i) & 7);
      word = (word & ~((uint64_t)0xff << sh)) | (byte << sh);
    }
  }
  (void)hipFree(d);
  return rc;
}

// Fill-reducing ordering of the cameras inside the reduced camera system -- what `perm` asks of the reference's sparse
// LDL' (amd(A) / Metis.permutation, src/lm.jl:84-88, consumed by ldl_analyse, src/ldl_aux.jl:246-283).  The tile pattern of S
// depends on how the cameras are numbered; a BAL file promises nothing about that.  Cameras keep their numbers everywhere
// else (x, J, Hcc, gc): only S, its right-hand side and its solution live in the new order (SchurTasks::cam_of / pos).
// pos_out: block row of every camera, empty when the caller's numbering stays.
static int order_cameras(ba_problem *p, LMWorkFull *w, std::vector<int> *pos_out) {
  pos_out->clear();
  w->h_cam_perm.clear();
  w->order_name = "natural";
  w->order_split = 0;
  int method = w->order_method;
  if (const char *e = getenv("BA_CAM_ORDER")) {  // amd | metis | natural: overrides the handle's setting (experiments)
    method = e[0] == 'n' ? BA_ORDER_NATURAL : (e[0] == 'm' ? BA_ORDER_METIS : BA_ORDER_AMD);
  }
  const int64_t n = p->ncams;
  // (fewer cameras than two tiles hold: dense anyway; several ranks: with per-rank ownership of S only, and at most 63 ranks
  // -- the graph's bits are summed as 6-bit counts)
  if (method == BA_ORDER_NATURAL || n < 2 * (NB / 9)) return BA_OK;
  if (p->comm.active() && (!w->ldl.own_only || p->comm.world > 63)) return BA_OK;
  CamGraph g;
  cam_graph_build(n, p->npnts, p->h_pt_ptr.data(), p->h_pt_obs.data(), p->h_cam0.data(), &g);
  BA_CHECK(allreduce_cam_graph(p, &g));
  // more than half of all camera pairs share points: S is dense at tile granularity whatever the order
  if ((double)g.edges() > 0.25 * (double)n * (double)(n - 1)) return BA_OK;
  std::vector<int> perm;
  const char *name = "natural";
  int split = 0;
  cam_order(g, method, NB, &perm, &name, &split);
  bool identity = true;
  for (int64_t k = 0; k < n && identity; k++) identity = perm[(size_t)k] == (int)k;
  if (identity) return BA_OK;
  w->order_name = name;
  w->order_split = split;
  pos_out->resize((size_t)n);
  for (int64_t k = 0; k < n; k++) (*pos_out)[(size_t)perm[(size_t)k]] = (int)k;
  BA_CHECK(upload_vec(&w->tasks.cam_of, perm));
  BA_CHECK(upload_vec(&w->tasks.pos, *pos_out));
  w->h_cam_perm.swap(perm);
  return BA_OK;
}

// what only the direct solves need: the tiles of S, the Schur task list, the per-observation Y blocks
static int ensure_dense(ba_problem *p, LMWorkFull *w) {
  if (w->ldl.S) return BA_OK;
  BA_CHECK(dmalloc(&w->Yobs, 6 * p->nobs));
  std::vector<int> pos;
  BA_CHECK(order_cameras(p, w, &pos));
  BA_CHECK(build_tasks(p, &w->tasks, pos));
  // Block-sparse reduced camera system (one GPU): symbolic factorisation of the tile occupancy; the list schedule is used
  // when the pattern's trailing updates are at most 60 % of the dense factorisation's (BA_SPARSE_S=1 / 0 forces it on /
  // off).  Every camera pair sharing points (the default synthetic generator, small problems) gives flop_fill = 1: dense.
  // several ranks: a rank's keys only show the tiles ITS observations touch; the pattern is that of the sum
  if (dist_factor_on(p)) BA_CHECK(allreduce_flags(p, &w->tasks.tile_occ));
  tile_pattern_build(w->ldl.nt, w->tasks.tile_occ, &w->pattern, w->order_split);
  w->tasks.tile_occ.clear();
  w->tasks.tile_occ.shrink_to_fit();
  const char *e = getenv("BA_SPARSE_S");
  const bool want = e ? e[0] != '0' : (w->pattern.flop_fill <= 0.6 && w->ldl.nt >= 8);
  w->use_pattern = want && (!p->comm.active() || w->ldl.own_only);  // (several ranks: with per-rank ownership of S only)
  if (w->use_pattern) BA_CHECK(dense_ldl_use_pattern(&w->ldl, &w->pattern));  // (compressed layout: before S is allocated)
  BA_CHECK(dense_ldl_alloc_S(&w->ldl));
  if (w->ldl.own_only) BA_CHECK(build_chunks(p, w));
  return BA_OK;
}

void lm_free(ba_problem *p) {
  if (!p->lm) return;
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  dense_ldl_free(&w->ldl);
  if (w->have32) {
    dense_ldl_free(&w->ldl32);
    (void)hipFree(w->rhs32);
  }
  for (SchurChunk &c : w->chunks) {
    if (c.cco_alloc) (void)hipFree(c.cco_alloc);
    if (c.keys) (void)hipFree(c.keys);
    if (c.skeys) (void)hipFree(c.skeys);
  }
  for (int q = 0; q < 2; q++) {
    if (w->ev_stage_ready[q]) (void)hipEventDestroy(w->ev_stage_ready[q]);
    if (w->ev_stage_free[q]) (void)hipEventDestroy(w->ev_stage_free[q]);
  }
  if (w->stage) (void)hipFree(w->stage);
  if (w->stage32) (void)hipFree(w->stage32);
  if (w->xf) (void)hipFree(w->xf);
  if (w->rf) (void)hipFree(w->rf);
  if (w->Jf) (void)hipFree(w->Jf);
  void *ptrs[] = {w->jn2, w->dcol, w->damp, w->Jq, w->rq, w->x, w->x_trial, w->delta, w->r, w->r_trial, w->J, w->Hpp, w->gp, w->Uinv, w->u, w->Hcc,
                  w->partial, w->partial_multi, w->colscale, w->Yobs, w->s.scal_rep, w->tasks.key_ptr, w->tasks.key_ca, w->tasks.key_cb,
                  w->tasks.task_a, w->tasks.task_b, w->tasks.skey, w->tasks.skey_c0, w->tasks.chunk_t0,
                  w->tasks.chunk_t1, w->tasks.partial, w->tasks.cam_of, w->tasks.pos, w->s.own_red ? w->s.red : nullptr};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  for (int q = 0; q < 2; q++) {
    if (w->g_step[q]) (void)hipGraphExecDestroy(w->g_step[q]);
    if (w->g_refresh[q]) (void)hipGraphExecDestroy(w->g_refresh[q]);
  }
  for (double *q : {w->cgx, w->cgr, w->cgz, w->cgp, w->cgq, w->cgt, w->cgh, w->zero3, w->blk45, w->cg_scal})
    if (q) (void)hipFree(q);
  if (w->h_cg) (void)hipHostFree(w->h_cg);
  if (w->cam_pnt) (void)hipFree(w->cam_pnt);
  if (w->d_lambda) (void)hipFree(w->d_lambda);
  if (w->h_lambda) (void)hipHostFree(w->h_lambda);
  if (w->h_flag) (void)hipHostFree(w->h_flag);
  if (w->s.h_sh) (void)hipHostFree(w->s.h_sh);
  if (w->s.h_rp) (void)hipHostFree(w->s.h_rp);
  delete w;
  p->lm = nullptr;
}

// all-reduce [off, off+count) of the reduce buffer over the ranks (no-op without a communicator; a communicator of one
// rank is still called: lets one GPU exercise the path)
static int comm_sum(ba_problem *p, LMWorkFull *w, int64_t off, int64_t count, hipStream_t st) {
  return comm_allreduce(p, w->s.red + off, count, st);
}

// the partial sums of S held by every rank -> the complete tile columns on their owners (distributed factorisation), or
// the complete S everywhere (replicated); the right-hand side is needed by every rank either way
// s32: the factorisation will run in Float32 and nothing needs the Float64 sum (no column scaling): every rank rounds its
// partial sums to Float32 first and the owners receive Float32 sums -- half the bytes of the largest transfer of the
// iteration (Final-13682: 30 GB instead of 60); the sum of `world` rounded partials differs from the rounded sum by a few
// Float32 ulps, the level of the factorisation itself.
static int reduce_camera_system(ba_problem *p, LMWorkFull *w, hipStream_t st, bool s32 = false) {
  if (!p->comm.active()) return BA_OK;
  if (!dist_factor_on(p)) {  // replicated: the whole S and the right-hand side everywhere
    BA_CHECK(comm_allreduce(p, w->ldl.S, w->ldl.s_tiles * NB * NB, st));
    return comm_sum(p, w, w->s.off_rhs, w->npad, st);
  }
  if (s32) BA_CHECK(launch_convert(w->ldl.S, w->ldl32.S, w->ldl.s_tiles * NB * NB, st));
  BA_CHECK(comm_group_begin(p));
  int rc = BA_OK;
  for (int r = 0; r < w->ldl.world && rc == BA_OK; r++) {
    const int64_t b = w->ldl.own_range[(size_t)r], e = w->ldl.own_range[(size_t)r + 1];
    rc = s32 ? comm_reduce_f32(p, w->ldl32.S + b * NB * NB, (e - b) * NB * NB, r, st)
             : comm_reduce(p, w->ldl.S + b * NB * NB, (e - b) * NB * NB, r, st);
  }
  BA_CHECK(comm_group_end(p));
  BA_CHECK(rc);
  return comm_sum(p, w, w->s.off_rhs, w->npad, st);
}

// r, J and the normal-equation blocks at w->x; fills sharded/replicated scalars RSQ?, GP, GC, X_P, X_C
// publish: the last reduction kernel also writes the controller's scalars to the pinned host buffers (recorded sequences)
static int refresh_linearisation(ba_problem *p, LMWorkFull *w, bool residual_too, hipStream_t st, bool xf32 = false, bool publish = false) {
  if (xf32) {  // w->x holds Float32 values: evaluate with the Float32 kernels, widen (exact)
    BA_CHECK(launch_convert(w->x, w->xf, w->nvar, st));
    if (residual_too) {
      BA_CHECK(launch_residual_f32(p, w->xf, w->rf, st));
      BA_CHECK(launch_convert(w->rf, w->r, w->nequ, st));
    }
  } else if (residual_too) {
    BA_CHECK(launch_residual_f64(p, w->x, w->r, st));
  }
  if (xf32) {
    BA_CHECK(launch_jac_coord_f32(p, w->xf, w->Jf, st));
    BA_CHECK(launch_convert(w->Jf, w->J, 24 * p->nobs, st));
  } else {
    BA_CHECK(launch_jac_coord_f64(p, w->x, w->J, st));
  }
  BA_CHECK(launch_point_blocks(p, w->J, w->r, w->Hpp, w->gp, st));
  BA_CHECK(launch_cam_blocks(p, w->J, w->r, w->Hcc, w->gc, st));
  // gc, the diagonal of the camera block (the column scalings need the global one) and the linearisation scalars are
  // adjacent in the reduce buffer: one all-reduce
  BA_CHECK(launch_hcc_diag(p, w->Hcc, w->hdiag, st));
  // |r|^2, |gp|^2, |x_points|^2 and -- of the all-reduced gc -- |gc|^2, |x_cameras|^2: one launch pair on one rank, two with a
  // communicator (the camera sums wait for the all-reduce); bit-identical to launch_sumsq per vector either way
  SumsqJobs jobs;
  jobs.add(w->r, w->nequ, w->scal, SH_RSQ);
  jobs.add(w->gp, 3 * p->npnts, w->scal, SH_GP);
  jobs.add(w->x, 3 * p->npnts, w->scal, SH_X_P);
  if (p->comm.active()) {
    BA_CHECK(launch_sumsq_multi(p, &jobs, w->partial_multi, st));
    BA_CHECK(comm_sum(p, w, w->s.off_gc, 2 * w->npad + SH_LIN_COUNT, st));
    jobs = SumsqJobs();
  }
  if (w->f16) BA_CHECK(launch_col_sq(p, w->Hpp, w->hdiag, w->jn2, st));  // |J_j|^2 before the blocks are overwritten by scaled ones
  jobs.add(w->gc, w->n, w->s.scal_rep, RP_GC);
  jobs.add(w->x + 3 * p->npnts, w->n, w->s.scal_rep, RP_X_C);
  if (publish) jobs.publish(w->scal, SH_COUNT, w->s.h_sh, w->s.scal_rep, RP_COUNT, w->s.h_rp, nullptr, nullptr);
  BA_CHECK(launch_sumsq_multi(p, &jobs, w->partial_multi, st));
  return BA_OK;
}

static int fetch_scalars(ba_problem *p, LMWorkFull *w, hipStream_t st) {
  BA_CHECK(launch_publish(w->scal, SH_COUNT, w->s.h_sh, w->s.scal_rep, RP_COUNT, w->s.h_rp, nullptr, nullptr, st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  return BA_OK;
}

// ---- facto = PCG ------------------------------------------------------------------------------------------------------
// The reference has no iterative branch (its solves are sparse direct, src/lm.jl:61-112); SURVEY 8(f) lists PCG on the
// reduced camera system as the way to Final-scale problems, where the dense S (60 GB for Final-13682) and its n^3/3 flops
// stop paying.  Here S = Hcc + lambda I - W U^-1 W' is applied, never formed: two sweeps over J per product (the
// back-substitution pass by point, the right-hand-side pass by camera), preconditioned by its 9 x 9 diagonal blocks.  On
// several ranks the product is summed by ONE all-reduce of 9 ncams doubles per CG iteration; everything else is replicated.
static int ensure_pcg(ba_problem *p, LMWorkFull *w) {
  if (w->cgx) return BA_OK;
  auto dm = [](double **q, int64_t cnt) -> int {
    BA_HIP_CHECK(hipMalloc((void **)q, (size_t)(cnt > 0 ? cnt : 1) * sizeof(double)));
    BA_HIP_CHECK(hipMemset(*q, 0, (size_t)(cnt > 0 ? cnt : 1) * sizeof(double)));
    return BA_OK;
  };
  for (double **q : {&w->cgx, &w->cgr, &w->cgz, &w->cgp, &w->cgq, &w->cgt}) BA_CHECK(dm(q, w->npad));
  BA_CHECK(dm(&w->cgh, 3 * p->npnts));
  BA_CHECK(dm(&w->zero3, 3 * p->npnts));
  BA_CHECK(dm(&w->blk45, 45 * p->ncams));
  BA_CHECK(dm(&w->cg_scal, 8));
  BA_HIP_CHECK(hipDeviceSynchronize());  // (the null-stream memsets above are not ordered against the handle's stream)
  BA_HIP_CHECK(hipHostMalloc((void **)&w->h_cg, 8 * sizeof(double)));
  return BA_OK;
}

// q = S v: the point sweep, then the camera sweep (which adds Hcc v and, on one rank, the damping); on several ranks the
// partial products are summed by one all-reduce and every rank adds the damping to the full sum
static int pcg_matvec(ba_problem *p, LMWorkFull *w, double lambda, const double *v, double *q, hipStream_t st) {
  const bool shared = p->comm.active();
  if (p->point_sorted) BA_CHECK(launch_wtv(p, w->J, w->Uinv, v, w->cgh, st));  // h = -U^-1 W' v
  else BA_CHECK(launch_backsub(p, w->J, w->Uinv, w->zero3, v, w->cgh, st));  // (observations not grouped by point)
  BA_CHECK(launch_wuw(p, w->J, w->cgh, w->Hcc, v, shared ? 0.0 : lambda, q, st, w->cam_pnt));
  if (!shared) return BA_OK;
  BA_CHECK(comm_allreduce(p, q, w->n, st));
  return launch_axpy_s(p, w->n, lambda, v, q, st);
}

static int pcg_fetch(LMWorkFull *w, hipStream_t st) {
  BA_HIP_CHECK(hipMemcpyAsync(w->h_cg, w->cg_scal, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  return BA_OK;
}

// S x = rhs (w->rhs in, solution out) to |r| <= pcg_tol |rhs| or pcg_maxit iterations.  alpha and beta are formed on the
// device (k_cg_alpha, k_cg_beta_dir); the host reads the scalars once per iteration, for the stopping test.
static int pcg_solve(ba_problem *p, LMWorkFull *w, double lambda, hipStream_t st) {
  BA_CHECK(ensure_pcg(p, w));
  const int64_t n = w->n;
  const int maxit = w->pcg_maxit > 0 ? w->pcg_maxit : 1000;
  BA_HIP_CHECK(hipMemsetAsync(w->ldl.flag, 0, sizeof(int), st));
  BA_CHECK(launch_schur_diag(p, w->J, w->Uinv, w->Hcc, w->blk45, st));
  BA_CHECK(comm_allreduce(p, w->blk45, 45 * p->ncams, st));
  BA_CHECK(launch_pcg_factor(p, lambda, w->blk45, w->ldl.flag, st));  // a block that is not positive definite -> SQDException
  BA_HIP_CHECK(hipMemsetAsync(w->cgx, 0, (size_t)n * sizeof(double), st));
  BA_HIP_CHECK(hipMemcpyAsync(w->cgr, w->rhs, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
  BA_CHECK(launch_cg_step(p, w->cg_scal, w->blk45, w->cgp, w->cgq, w->cgx, w->cgr, w->cgz, w->cgt, 1, st));  // z, r.r, r.z
  BA_CHECK(launch_cg_beta_dir(p, n, w->cgt, w->cg_scal, w->cgz, w->cgp, 1, st));                              // p = z
  BA_CHECK(pcg_fetch(w, st));
  const double b2 = w->h_cg[1];
  int it = 0;
  if (b2 > 0 && w->h_cg[2] == w->h_cg[2]) {
    for (it = 1; it <= maxit; it++) {
      BA_CHECK(pcg_matvec(p, w, lambda, w->cgp, w->cgq, st));
      BA_CHECK(launch_cg_alpha(p, n, w->cgp, w->cgq, w->cg_scal, st));
      BA_CHECK(launch_cg_step(p, w->cg_scal, w->blk45, w->cgp, w->cgq, w->cgx, w->cgr, w->cgz, w->cgt, 0, st));
      BA_CHECK(launch_cg_beta_dir(p, n, w->cgt, w->cg_scal, w->cgz, w->cgp, 0, st));
      BA_CHECK(pcg_fetch(w, st));
      const double pq = w->h_cg[0], rr = w->h_cg[1];
      // p.q <= 0: S not positive definite along p (or NaN) -- alpha was 0, nothing moved; keep what there is, the LM test
      // judges the step
      if (!(pq > 0) || !(rr == rr) || rr <= w->pcg_tol * w->pcg_tol * b2) break;
    }
  }
  w->n_cg += it < maxit ? it : maxit;
  BA_HIP_CHECK(hipMemcpyAsync(w->rhs, w->cgx, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
  return BA_OK;
}

// delta = -(J'J + lambda I)^-1 J'r at the current linearisation; also |J delta + r|^2 -> SH_MODEL, |delta|^2
// h_lambda (recorded sequences): pinned host scalar the first kernel copies into d_lambda
static int linear_step(ba_problem *p, LMWorkFull *w, double lambda, int normalize, hipStream_t st,
                       bool facto_f32 = false, const double *d_lambda = nullptr, const double *h_lambda = nullptr) {
  // d_lambda: the damping is read from device memory (recorded launches); `lambda` is then the multiplier 1
  // every rank holds partial Hcc / Schur sums; the lambda I of the camera block is added by rank 0 only
  const double lam_diag = (p->rank == 0) ? lambda : 0.0;
  const double *Jl = w->J, *rl = w->r, *damp = nullptr;
  constexpr double MU16 = 0.1 * 6.55e4;  // lma_aux.jl:44-48
  if (w->f16) {
    // facto_type = Float16: columns scaled by their norms, entries and right-hand side rounded to Float16 (k_f16_cols); the
    // normal-equation blocks are rebuilt from the rounded copies, the damping is per column
    BA_CHECK(launch_f16_scale(p, lambda, MU16, w->jn2, w->J, w->r, w->dcol, w->damp, w->Jq, w->rq, st));
    BA_CHECK(launch_point_blocks(p, w->Jq, w->rq, w->Hpp, w->gp, st));
    BA_CHECK(launch_cam_blocks(p, w->Jq, w->rq, w->Hcc, w->gc, st));
    Jl = w->Jq;
    rl = w->rq;
    damp = w->damp;
    normalize = 0;  // lm.jl:156,232: no column scaling of J in the Float16 branch
  }
  // (single-rank direct path: the right-hand side buffer is cleared by this kernel instead of a memset node of its own)
  const bool rhs_here = !w->pcg && !w->ldl.own_only;
  // recorded sequences read the damping from pinned host memory (h_lambda).  In k_schur_prep every wave would fetch that
  // scalar over PCIe -- unnoticeable for the few thousand waves of a small problem, where the saved copy node pays, but 15 k
  // waves on the Venice shape and 70 k on Final-13682's: there one thread copies it to device memory first
  if (h_lambda && p->npnts > 200000) {
    BA_CHECK(launch_publish(h_lambda, 1, const_cast<double *>(d_lambda), nullptr, 0, nullptr, nullptr, nullptr, st));
    h_lambda = nullptr;
  }
  BA_CHECK(launch_schur_prep(p, lambda, w->Hpp, w->gp, w->Uinv, w->u, st, h_lambda ? h_lambda : d_lambda, damp,
                             h_lambda ? const_cast<double *>(d_lambda) : nullptr, rhs_here ? w->rhs : nullptr, rhs_here ? w->npad : 0));
  if (w->pcg) {  // the reduced camera system is applied, not formed (pcg_solve); no column scaling: block Jacobi has its own
    BA_HIP_CHECK(hipMemsetAsync(w->rhs, 0, (size_t)w->npad * sizeof(double), st));
    BA_CHECK(launch_schur_rhs(p, Jl, rl, w->u, w->rhs, st, w->cam_pnt));
    BA_CHECK(comm_sum(p, w, w->s.off_rhs, w->npad, st));
    w->last_f32 = false;
    BA_CHECK(pcg_solve(p, w, lambda, st));
    double *dcp = w->delta + 3 * p->npnts;
    BA_HIP_CHECK(hipMemcpyAsync(dcp, w->rhs, (size_t)w->n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return launch_backsub(p, Jl, w->Uinv, w->u, dcp, w->delta, st, rl, w->cr0(), w->partial, w->scal, SH_MODEL, &w->model_done);
  }
  const bool dist = dist_factor_on(p);
  const bool reduce32 = dist && facto_f32 && normalize == 0;
  if (reduce32) BA_CHECK(ensure_f32(w));
  if (w->ldl.own_only) {
    // per-rank ownership of S: chunk by chunk -- this rank's share of the chunk's sums goes straight into its own tiles
    // when it owns the chunk, else into the staging buffer -- and every chunk is reduced onto its owner at once
    BA_CHECK(launch_schur_pre(p, &w->tasks, Jl, w->Uinv, w->Yobs, st));
    if (w->assembly_rs) {
      // reduce-scatter form (build_chunks_rs): every chunk is one slice of EVERY owner's columns, assembled into a staging
      // buffer of `world` segments and reduce-scattered in place; the owner copies its segment into its tiles.  With two
      // buffers the transfer of chunk c (transfer stream) runs beside the assembly of chunk c+1 (main stream).
      const int bufs = w->stage_bufs, me = w->ldl.rank;
      const int64_t buf_elems = (w->stage_tiles / bufs) * NB * NB;
      hipStream_t ts = bufs == 2 ? w->ldl.hoist : st;
      bool used[2] = {false, false};
      int ci = 0;
      for (const SchurChunk &c : w->chunks) {
        const int b = bufs == 2 ? (ci++ & 1) : 0;
        double *dst = w->stage + b * buf_elems;
        if (bufs == 2 && used[b]) BA_HIP_CHECK(hipStreamWaitEvent(st, w->ev_stage_free[b], 0));  // its last transfer is through
        BA_CHECK(launch_schur_chunk(p, &w->tasks, &c, Jl, w->Yobs, w->Hcc, lam_diag, dst, w->n, p->rank == 0 ? w->npad : w->n, st,
                                    d_lambda, damp));
        const int64_t seg_elems = c.seg * NB * NB, my_elems = c.my_n * NB * NB;
        float *d32 = reduce32 ? w->stage32 + b * buf_elems : nullptr;
        if (reduce32) BA_CHECK(launch_convert(dst, d32, c.ntiles * NB * NB, st));
        if (bufs == 2) {
          BA_HIP_CHECK(hipEventRecord(w->ev_stage_ready[b], st));
          BA_HIP_CHECK(hipStreamWaitEvent(ts, w->ev_stage_ready[b], 0));
        }
        BA_CHECK(comm_reduce_scatter(p, reduce32 ? (void *)d32 : (void *)dst, seg_elems, reduce32, ts));
        if (my_elems > 0) {
          if (reduce32)
            BA_HIP_CHECK(hipMemcpyAsync(w->ldl32.S + c.my_t0 * NB * NB, d32 + (int64_t)me * seg_elems, (size_t)my_elems * sizeof(float),
                                        hipMemcpyDeviceToDevice, ts));
          else
            BA_HIP_CHECK(hipMemcpyAsync(w->ldl.S + c.my_t0 * NB * NB, dst + (int64_t)me * seg_elems, (size_t)my_elems * sizeof(double),
                                        hipMemcpyDeviceToDevice, ts));
        }
        if (bufs == 2) {
          BA_HIP_CHECK(hipEventRecord(w->ev_stage_free[b], ts));
          used[b] = true;
        }
      }
      for (int b = 0; b < 2; b++)
        if (used[b]) BA_HIP_CHECK(hipStreamWaitEvent(st, w->ev_stage_free[b], 0));
    } else
    for (const SchurChunk &c : w->chunks) {
      const bool mine = c.owner == w->ldl.rank;
      double *dest = mine ? w->ldl.S + c.t0 * NB * NB : w->stage;
      BA_CHECK(launch_schur_chunk(p, &w->tasks, &c, Jl, w->Yobs, w->Hcc, lam_diag, dest, w->n, p->rank == 0 ? w->npad : w->n, st,
                                  d_lambda, damp));
      if (reduce32) {  // Float32 factorisation without column scaling: the partial sums travel as Float32
        float *d32 = mine ? w->ldl32.S + c.t0 * NB * NB : w->stage32;
        BA_CHECK(launch_convert(dest, d32, c.ntiles * NB * NB, st));
        BA_CHECK(comm_reduce_f32(p, d32, c.ntiles * NB * NB, c.owner, st));
      } else {
        BA_CHECK(comm_reduce(p, dest, c.ntiles * NB * NB, c.owner, st));
      }
    }
    BA_HIP_CHECK(hipMemsetAsync(w->rhs, 0, (size_t)w->npad * sizeof(double), st));
    BA_CHECK(launch_schur_rhs(p, Jl, rl, w->u, w->rhs, st, w->cam_pnt, w->tasks.pos));
    BA_CHECK(comm_sum(p, w, w->s.off_rhs, w->npad, st));
  } else {
    BA_CHECK(launch_schur_blocks(p, &w->tasks, Jl, w->Uinv, w->Yobs, w->Hcc, lam_diag, w->ldl.S, w->ldl.col_off, w->n,
                                 p->rank == 0 ? w->npad : w->n, st, d_lambda, damp, w->ldl.s_tiles));
    BA_CHECK(launch_schur_rhs(p, Jl, rl, w->u, w->rhs, st, w->cam_pnt, w->tasks.pos));
    BA_CHECK(reduce_camera_system(p, w, st, reduce32));
  }
  if (normalize != 0) {  // :J / :A column scaling of the camera system from the GLOBAL diagonal (refresh_linearisation)
    BA_CHECK(launch_cam_scale(p, w->hdiag, normalize == 2 ? lambda : 0.0, w->colscale, st, d_lambda, w->tasks.pos));
    if (w->ldl.own_only && w->use_pattern)
      BA_CHECK(launch_scale_S_list(p, w->n, w->colscale, w->ldl.S, w->ldl.own_tiles,
                                   w->ldl.own_range[(size_t)w->ldl.rank + 1] - w->ldl.own_range[(size_t)w->ldl.rank], st));
    else if (w->ldl.own_only)
      BA_CHECK(launch_scale_S_own(p, w->n, w->colscale, w->ldl.S, w->ldl.col_off, w->ldl.own_cols, w->ldl.own_pref,
                                  (int)w->ldl.h_own_cols.size(), w->ldl.own_range[(size_t)w->ldl.rank + 1] - w->ldl.own_range[(size_t)w->ldl.rank], st));
    else
      BA_CHECK(launch_scale_S(p, w->n, w->ldl.nt, w->colscale, w->ldl.S, w->ldl.col_off, st));
    BA_CHECK(launch_scale_vec(p, w->n, w->colscale, w->rhs, 1, st));
  }
  w->last_f32 = facto_f32;
  if (facto_f32) {  // round the assembled system to Float32, factor and solve there, widen the solution
    BA_CHECK(ensure_f32(w));
    if (!reduce32) BA_CHECK(launch_convert(w->ldl.S, w->ldl32.S, w->ldl.s_tiles * NB * NB, st));
    BA_CHECK(launch_convert(w->rhs, w->rhs32, w->npad, st));
    if (dist) {
      BA_CHECK(dense_ldl_factor_dist<float>(p, &w->ldl32, st, w->ldl32.own_only ? w->rhs32 : (float *)nullptr));
      BA_CHECK(dense_ldl_solve<float>(p, &w->ldl32, w->rhs32, st, w->ldl32.own_only));
    } else {
      BA_CHECK(dense_ldl_factor<float>(p, &w->ldl32, st, nullptr, w->rhs32));
      BA_CHECK(dense_ldl_solve<float>(p, &w->ldl32, w->rhs32, st, true));
    }
    BA_CHECK(launch_convert(w->rhs32, w->rhs, w->npad, st));
  } else if (dist) {
    BA_CHECK(dense_ldl_factor_dist(p, &w->ldl, st, w->ldl.own_only ? w->rhs : (double *)nullptr));
    BA_CHECK(dense_ldl_solve(p, &w->ldl, w->rhs, st, w->ldl.own_only));
  } else {
    BA_CHECK(dense_ldl_factor(p, &w->ldl, st, nullptr, w->rhs));  // forward substitution of rhs rides along
    BA_CHECK(dense_ldl_solve(p, &w->ldl, w->rhs, st, true));
  }
  double *dc = w->delta + 3 * p->npnts;
  if (normalize != 0) BA_CHECK(launch_scale_vec(p, w->n, w->colscale, w->rhs, 1, st));  // dc = D^-1 dc'
  if (w->tasks.pos) BA_CHECK(launch_gather_cams(p, w->tasks.pos, w->rhs, dc, st));  // block rows of S -> camera order
  else BA_HIP_CHECK(hipMemcpyAsync(dc, w->rhs, (size_t)w->n * sizeof(double), hipMemcpyDeviceToDevice, st));
  // the model value of the step rides along with the back-substitution (not in the Float16 branch: its step is rescaled below)
  BA_CHECK(launch_backsub(p, Jl, w->Uinv, w->u, dc, w->delta, st, w->f16 ? nullptr : rl, w->cr0(), w->partial, w->scal, SH_MODEL,
                          &w->model_done));
  if (w->f16) BA_CHECK(launch_scale_scalar(p, w->nvar, w->delta, 1.0 / MU16, st));  // the right-hand side was -Jh' r / mu
  return BA_OK;
}

// cr: the model value is |J delta + cr r|^2 (1 outside the line search)
// defer_delta: |delta_points|^2 and |delta_cameras|^2 are left to trial_point (one reduction pair with |r_trial|^2)
static int step_scalars(ba_problem *p, LMWorkFull *w, hipStream_t st, double cr = -1.0, bool defer_delta = false) {
  const bool first = cr < 0;  // the step as linear_step left it (the line search calls with a rescaled delta and its own cr)
  if (first) cr = w->cr0();
  if (!(first && w->model_done)) BA_CHECK(launch_model_sq(p, w->J_lin(), w->r_lin(), w->delta, w->partial, w->scal, SH_MODEL, st, cr));
  w->model_done = false;
  if (!defer_delta) {
    SumsqJobs jobs;
    jobs.add(w->delta, 3 * p->npnts, w->scal, SH_DELTA_P);
    jobs.add(w->delta + 3 * p->npnts, w->n, w->s.scal_rep, RP_DELTA_C);
    BA_CHECK(launch_sumsq_multi(p, &jobs, w->partial_multi, st));
  }
  return BA_OK;
}

// with_delta: the step's two norms ride with |r_trial|^2 (step_scalars was told to leave them); publish_flag: the reduction
// also writes the controller's scalars and this pivot flag to the pinned host buffers (recorded sequences)
static int trial_point(ba_problem *p, LMWorkFull *w, hipStream_t st, bool xf32 = false, bool with_delta = false,
                       const int *publish_flag = nullptr) {
  BA_CHECK(launch_axpy(p, w->nvar, w->x, w->delta, w->x_trial, st));
  if (xf32) {  // x_suiv is a Float32 vector in the reference: round, evaluate in Float32
    BA_CHECK(launch_convert(w->x_trial, w->xf, w->nvar, st));
    BA_CHECK(launch_convert(w->xf, w->x_trial, w->nvar, st));
    BA_CHECK(launch_residual_f32(p, w->xf, w->rf, st));
    BA_CHECK(launch_convert(w->rf, w->r_trial, w->nequ, st));
  } else {
    BA_CHECK(launch_residual_f64(p, w->x_trial, w->r_trial, st));
  }
  if (!with_delta) return launch_sumsq(p, w->nequ, w->r_trial, w->partial, w->scal, SH_RSQ_TRIAL, st);
  SumsqJobs jobs;
  jobs.add(w->delta, 3 * p->npnts, w->scal, SH_DELTA_P);
  jobs.add(w->delta + 3 * p->npnts, w->n, w->s.scal_rep, RP_DELTA_C);
  jobs.add(w->r_trial, w->nequ, w->scal, SH_RSQ_TRIAL);
  if (publish_flag) jobs.publish(w->scal, SH_COUNT, w->s.h_sh, w->s.scal_rep, RP_COUNT, w->s.h_rp, publish_flag, w->h_flag);
  return launch_sumsq_multi(p, &jobs, w->partial_multi, st);
}

static int check_pivot(ba_problem *p, LMWorkFull *w, hipStream_t st) {
  int h = 0;
  BA_HIP_CHECK(hipMemcpyAsync(&h, w->last_f32 ? w->ldl32.flag : w->ldl.flag, sizeof(int), hipMemcpyDeviceToHost, st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  if (h == 2) {
    ba_set_error("dense factorisation: a hoisted diagonal tile never became ready (internal scheduling error)");
    return BA_ERR_HIP;
  }
  if (h) {
    ba_set_error("reduced camera system: exactly zero pivot (SQDException in the reference)");
    return BA_ERR_ZERO_PIVOT;
  }
  return BA_OK;
}

// ---- recorded launch sequences ------------------------------------------------------------------------------------------
static bool graphs_allowed(ba_problem *p, LMWorkFull *w) {
  if (w->g_off || p->prof_on || p->comm.active() || w->f16 || w->pcg) return false;  // per-kernel events / communicator / Float16 path
  // the hoisted-diagonal schedule of large factorisations has a kernel wait for a flag raised by a kernel running
  // beside it: only with real streams is that concurrency certain (and the graphs gain nothing at that size)
  // (the block-sparse list schedule never hoists and is bound by its chain of short launches: recorded at any size)
  if (w->ldl.nt >= 34 && !w->use_pattern) return false;  // = HOIST_MIN_TILES + 2 of dense_ldl_factor (above its HOIST_MAX_TILES graphs gain nothing either)
  const char *e = getenv("BA_LM_GRAPH");
  return !(e && e[0] == '0');
}

template <typename F>
static int record_graph(hipStream_t st, hipGraphExec_t *out, F body) {
  hipGraph_t g = nullptr;
  BA_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  int rc = body();
  hipError_t e = hipStreamEndCapture(st, &g);
  if (rc != BA_OK) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  BA_HIP_CHECK(e);
  e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  BA_HIP_CHECK(e);
  return BA_OK;
}

// A hoisted diagonal kernel of the dense factorisation gave up waiting for its flag: the kernels were not running side by
// side (a counter-collecting profiler serialises them).  Switch the handle to the in-order schedule; the caller redoes
// the step (S was consumed by the abandoned factorisation).
static bool hoist_gave_up(LMWorkFull *w) {
  if (*w->h_flag != 2 || (w->ldl.hoist_disabled && w->ldl32.hoist_disabled)) return false;
  w->ldl.hoist_disabled = w->ldl32.hoist_disabled = true;
  return true;
}

// one trial step at damping `lambda`: linear solve, model decrease, trial residual, scalars and pivot flag to the host
static int trial_step(ba_problem *p, LMWorkFull *w, double lambda, int normalize, bool facto_f32, bool xf32,
                      hipStream_t st) {
  if (!graphs_allowed(p, w)) {
    BA_CHECK(linear_step(p, w, lambda, normalize, st, facto_f32));
    BA_CHECK(step_scalars(p, w, st, -1.0, true));
    BA_CHECK(trial_point(p, w, st, xf32, true));
    BA_CHECK(comm_sum(p, w, w->s.off_scal + SH_TRIAL_FIRST, SH_TRIAL_COUNT, st));
    BA_HIP_CHECK(hipMemcpyAsync(w->h_flag, w->last_f32 ? w->ldl32.flag : w->ldl.flag, sizeof(int), hipMemcpyDeviceToHost, st));
    BA_CHECK(fetch_scalars(p, w, st));  // (synchronises st)
    if (hoist_gave_up(w)) return trial_step(p, w, lambda, normalize, facto_f32, xf32, st);
    return BA_OK;
  }
  const int key = normalize + 4 * (facto_f32 ? 1 : 0) + 8 * (xf32 ? 1 : 0);
  if (w->g_key != key) {
    for (int q = 0; q < 2; q++) {
      if (w->g_step[q]) (void)hipGraphExecDestroy(w->g_step[q]);
      if (w->g_refresh[q]) (void)hipGraphExecDestroy(w->g_refresh[q]);
      w->g_step[q] = w->g_refresh[q] = nullptr;
    }
    w->g_key = key;
  }
  if (facto_f32) BA_CHECK(ensure_f32(w));  // no allocation while recording
  hipGraphExec_t &g = w->g_step[w->parity];
  if (!g) {
    const int grc = record_graph(st, &g, [&]() -> int {
      BA_CHECK(linear_step(p, w, 1.0, normalize, st, facto_f32, w->d_lambda, w->h_lambda));
      BA_CHECK(step_scalars(p, w, st, -1.0, true));
      BA_CHECK(trial_point(p, w, st, xf32, true, facto_f32 ? w->ldl32.flag : w->ldl.flag));  // (+ scalars and flag to the host)
      return BA_OK;
    });
    if (grc != BA_OK) {  // recording is an optimisation: without it the same launches are issued one by one
      (void)hipGetLastError();
      g = nullptr;
      w->g_off = true;
      return trial_step(p, w, lambda, normalize, facto_f32, xf32, st);
    }
  }
  *w->h_lambda = lambda;
  w->last_f32 = facto_f32;
  BA_HIP_CHECK(hipGraphLaunch(g, st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  return BA_OK;  // (recorded sequences never hoist: graphs_allowed)
}

// after an accepted step (x/x_trial, r/r_trial already swapped): J, the normal-equation blocks, J'r, scalars to the host
static int accept_refresh(ba_problem *p, LMWorkFull *w, bool xf32, hipStream_t st) {
  if (!graphs_allowed(p, w)) {
    BA_CHECK(refresh_linearisation(p, w, false, st, xf32));
    return fetch_scalars(p, w, st);
  }
  hipGraphExec_t &g = w->g_refresh[w->parity];
  if (!g) {
    const int grc = record_graph(st, &g, [&]() -> int {
      BA_CHECK(refresh_linearisation(p, w, false, st, xf32, true));  // (+ scalars to the host)
      return BA_OK;
    });
    if (grc != BA_OK) {
      (void)hipGetLastError();
      g = nullptr;
      w->g_off = true;
      return accept_refresh(p, w, xf32, st);
    }
  }
  BA_HIP_CHECK(hipGraphLaunch(g, st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  return BA_OK;
}

// Recorded sequences only: the refresh after an accepted step and the NEXT trial step (its damping is known at the moment
// of acceptance) are submitted back to back and waited for once.  The host's work between two launches -- waking up from
// the wait, the accept test, submitting a sequence of ~110 nodes -- otherwise leaves the device idle twice per iteration
// (Dubrovnik: 31 + 79 us of a 3.0 ms iteration, LadyBug: the same of 0.44 ms; rocprofv3 kernel trace).  The refresh and the
// trial write disjoint scalar slots, so one read of the pinned buffers serves both.  If the stopping tests that need the
// refreshed |J'r| or |x| then end the loop, the prefetched step is dropped (never counted, x untouched).
static bool can_prefetch_trial(ba_problem *p, LMWorkFull *w, int normalize, bool facto_f32, bool xf32) {
  const char *e = getenv("BA_LM_PREFETCH");  // read per call: a test compares both forms in one process
  if ((e && e[0] == '0') || !graphs_allowed(p, w)) return false;
  const int key = normalize + 4 * (facto_f32 ? 1 : 0) + 8 * (xf32 ? 1 : 0);
  return w->g_key == key && w->g_step[w->parity] && w->g_refresh[w->parity];
}
static int accept_refresh_and_trial(LMWorkFull *w, double lambda, bool facto_f32, hipStream_t st) {
  BA_HIP_CHECK(hipGraphLaunch(w->g_refresh[w->parity], st));
  *w->h_lambda = lambda;  // read by the trial sequence's first kernel; the previous trial sequence has completed
  w->last_f32 = facto_f32;
  BA_HIP_CHECK(hipGraphLaunch(w->g_step[w->parity], st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  return BA_OK;
}

static int lm_step_impl(ba_problem *p, const double *x, double lambda, double *delta, double *half_sq_model,
                        double *jtr, bool facto_f32, bool pcg = false, double tol = 0, int max_iter = 0, int *cg_iters = nullptr) {
  if (!p || !x || !delta) {
    ba_set_error("ba_lm_step: null argument");
    return BA_ERR_ARG;
  }
  BA_HIP_CHECK(hipSetDevice(p->device));
  BA_CHECK(lm_ensure(p));
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  hipStream_t st = p->stream;
  w->pcg = pcg;
  w->pcg_tol = tol > 0 ? tol : 1e-8;
  w->pcg_maxit = max_iter > 0 ? max_iter : 0;
  w->n_cg = 0;
  if (!pcg) BA_CHECK(ensure_dense(p, w));
  BA_HIP_CHECK(hipMemcpyAsync(w->x, x, (size_t)w->nvar * sizeof(double), hipMemcpyHostToDevice, st));
  w->f16 = false;
  BA_CHECK(refresh_linearisation(p, w, true, st));
  BA_CHECK(linear_step(p, w, lambda, 0, st, facto_f32));
  {  // same fallback as the LM loop: a hoisted diagonal kernel that gave up -> in-order schedule, redo the step
    BA_HIP_CHECK(hipMemcpyAsync(w->h_flag, facto_f32 ? w->ldl32.flag : w->ldl.flag, sizeof(int), hipMemcpyDeviceToHost, st));
    BA_HIP_CHECK(hipStreamSynchronize(st));
    if (hoist_gave_up(w)) BA_CHECK(linear_step(p, w, lambda, 0, st, facto_f32));
  }
  BA_CHECK(check_pivot(p, w, st));
  BA_CHECK(step_scalars(p, w, st));
  BA_CHECK(comm_sum(p, w, w->s.off_scal + SH_TRIAL_FIRST, SH_TRIAL_COUNT, st));
  BA_CHECK(fetch_scalars(p, w, st));
  BA_HIP_CHECK(hipMemcpyAsync(delta, w->delta, (size_t)w->nvar * sizeof(double), hipMemcpyDeviceToHost, st));
  if (jtr) {
    BA_HIP_CHECK(hipMemcpyAsync(jtr, w->gp, (size_t)3 * p->npnts * sizeof(double), hipMemcpyDeviceToHost, st));
    BA_HIP_CHECK(hipMemcpyAsync(jtr + 3 * p->npnts, w->gc, (size_t)w->n * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  BA_HIP_CHECK(hipStreamSynchronize(st));
  if (half_sq_model) *half_sq_model = 0.5 * w->s.h_sh[SH_MODEL];
  if (cg_iters) *cg_iters = (int)w->n_cg;
  w->pcg = false;
  return BA_OK;
}

extern "C" int ba_lm_schur_memory(ba_problem *p, int64_t *tiles_full, int64_t *tiles_held, int64_t *tiles_staging) {
  if (!p || !p->lm || !static_cast<LMWorkFull *>(p->lm)->ldl.S) {
    ba_set_error("ba_lm_schur_memory: no direct solve has run on this handle yet");
    return BA_ERR_ARG;
  }
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  if (tiles_full) *tiles_full = w->ldl.nt * (w->ldl.nt + 1) / 2;
  if (tiles_held) *tiles_held = w->ldl.s_tiles;
  if (tiles_staging) *tiles_staging = w->ldl.own_only ? w->stage_tiles : 0;
  return BA_OK;
}

extern "C" int ba_lm_schur_pattern(ba_problem *p, double *tile_fill, double *flop_fill, int *sparse_schedule) {
  if (!p || !p->lm) {
    ba_set_error("ba_lm_schur_pattern: no direct solve has run on this handle yet");
    return BA_ERR_ARG;
  }
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  if (!w->ldl.S) {
    ba_set_error("ba_lm_schur_pattern: no direct solve has run on this handle yet");
    return BA_ERR_ARG;
  }
  if (tile_fill) *tile_fill = w->pattern.tile_fill;
  if (flop_fill) *flop_fill = w->pattern.flop_fill;
  if (sparse_schedule) *sparse_schedule = w->use_pattern ? 1 : 0;
  return BA_OK;
}

// (re)select the camera ordering of the handle; structures built under another one are dropped
static int set_ordering(ba_problem *p, int method) {
  if (method < 0 || method > 2) {
    ba_set_error("camera ordering: 0 (:AMD), 1 (:Metis) or 2 (the caller's numbering)");
    return BA_ERR_ARG;
  }
  BA_CHECK(lm_ensure(p));
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  if (w->order_method == method) return BA_OK;
  if (w->ldl.S) {  // the task list, the pattern and the tiles depend on the order: start over
    lm_free(p);
    BA_CHECK(lm_ensure(p));
    w = static_cast<LMWorkFull *>(p->lm);
  }
  w->order_method = method;
  return BA_OK;
}

extern "C" int ba_lm_set_ordering(ba_problem *p, int method) {
  if (!p) {
    ba_set_error("ba_lm_set_ordering: null handle");
    return BA_ERR_ARG;
  }
  BA_HIP_CHECK(hipSetDevice(p->device));
  return set_ordering(p, method);
}

extern "C" int ba_lm_schur_ordering(ba_problem *p, int64_t *perm1, const char **name) {
  if (!p || !p->lm || !static_cast<LMWorkFull *>(p->lm)->ldl.S) {
    ba_set_error("ba_lm_schur_ordering: no direct solve has run on this handle yet");
    return BA_ERR_ARG;
  }
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  if (perm1)
    for (int64_t k = 0; k < p->ncams; k++) perm1[k] = (w->h_cam_perm.empty() ? k : (int64_t)w->h_cam_perm[(size_t)k]) + 1;
  if (name) *name = w->order_name;
  return BA_OK;
}

extern "C" int ba_lm_step_pcg(ba_problem *p, const double *x, double lambda, double tol, int max_iter, double *delta,
                              double *half_sq_model, double *jtr, int *cg_iters_out) {
  return lm_step_impl(p, x, lambda, delta, half_sq_model, jtr, false, true, tol, max_iter, cg_iters_out);
}

extern "C" int ba_lm_step(ba_problem *p, const double *x, double lambda, double *delta, double *half_sq_model,
                          double *jtr) {
  return lm_step_impl(p, x, lambda, delta, half_sq_model, jtr, false);
}

extern "C" int ba_lm_step_f32(ba_problem *p, const double *x, double lambda, double *delta, double *half_sq_model,
                              double *jtr) {
  return lm_step_impl(p, x, lambda, delta, half_sq_model, jtr, true);
}

// A scalar with the width Julia's promotion rules give it: an operation rounds to Float32 exactly when both operands are
// Float32 (Base promotion: Float32 op Float64 -> Float64).  Used by ba_lm_solve for Float32 models.
namespace ts {
struct TS {
  double v;
  int w;  // 32 | 64
};
static inline TS f32(double x) { return TS{(double)(float)x, 32}; }
static inline TS f64(double x) { return TS{x, 64}; }
static inline bool both32(TS a, TS b) { return a.w == 32 && b.w == 32; }
static inline TS add(TS a, TS b) { return both32(a, b) ? f32((float)a.v + (float)b.v) : f64(a.v + b.v); }
static inline TS sub(TS a, TS b) { return both32(a, b) ? f32((float)a.v - (float)b.v) : f64(a.v - b.v); }
static inline TS mul(TS a, TS b) { return both32(a, b) ? f32((float)a.v * (float)b.v) : f64(a.v * b.v); }
static inline TS div(TS a, TS b) { return both32(a, b) ? f32((float)a.v / (float)b.v) : f64(a.v / b.v); }
static inline TS max(TS a, TS b) { return TS{a.v > b.v ? a.v : b.v, both32(a, b) ? 32 : 64}; }
static inline TS sqrt(TS a) { return a.w == 32 ? f32(std::sqrt((float)a.v)) : f64(std::sqrt(a.v)); }
// x^n with a variable Int n (lm.jl:308,331).  The reference ran on Julia 1.3 / 1.4 (.travis.yml:6-9), whose
// ^(x::Float64, y::Integer) and ^(x::Float32, y::Integer) are llvm.pow of the exponent CONVERTED to the base's type
// (base/math.jl there; power_by_squaring is the generic fallback for other number types and only became the float path in
// Julia 1.8's pow_body): pow() / powf() of the rounded exponent is that operation.
static inline TS powi(TS a, int n) { return a.w == 32 ? f32(std::pow((float)a.v, (float)n)) : f64(std::pow(a.v, (double)n)); }
}  // namespace ts

static int lm_solve_impl(ba_problem *p, const ba_lm_opts *o, double *x_inout, bool x_on_device, ba_lm_stats *stats, ba_log_cb cb,
                         void *cb_ctx);

extern "C" int ba_lm_solve(ba_problem *p, const ba_lm_opts *o, double *x_inout, ba_lm_stats *stats, ba_log_cb cb,
                           void *cb_ctx) {
  return lm_solve_impl(p, o, x_inout, false, stats, cb, cb_ctx);
}

extern "C" int ba_lm_solve_dev(ba_problem *p, const ba_lm_opts *o, double *d_x_inout, ba_lm_stats *stats, ba_log_cb cb,
                               void *cb_ctx) {
  return lm_solve_impl(p, o, d_x_inout, true, stats, cb, cb_ctx);
}

static int lm_solve_impl(ba_problem *p, const ba_lm_opts *o, double *x_inout, bool x_on_device, ba_lm_stats *stats, ba_log_cb cb,
                         void *cb_ctx) {
  if (!p || !o || !x_inout || !stats) {
    ba_set_error("ba_lm_solve: null argument");
    return BA_ERR_ARG;
  }
  if (o->variant != 0 && o->variant != 1) {
    ba_set_error("ba_lm_solve: variant must be 0 (LevenbergMarquardt.jl) or 1 (lm.jl)");
    return BA_ERR_ARG;
  }
  if (o->facto < 0 || o->facto > 2) {
    ba_set_error("ba_lm_solve: facto must be 0 (:LDL), 1 (:QR) or 2 (:PCG)");
    return BA_ERR_ARG;
  }
  if (o->facto == 2 && o->facto_type == 2) {
    ba_set_error("ba_lm_solve: facto = :PCG runs in Float64 (facto_type = Float16 belongs to the :LDL branch)");
    return BA_ERR_ARG;
  }
  if (o->facto == 2 && o->normalize != 0) {
    ba_set_error("ba_lm_solve: facto = :PCG has its own scaling (block-Jacobi preconditioner): normalize must be :None");
    return BA_ERR_ARG;
  }
  if (o->facto == 2 && o->facto_type == 1 && !o->x_f32) {
    ba_set_error("ba_lm_solve: facto = :PCG runs in Float64; facto_type = Float32 belongs to the direct branches "
                 "(for a Float32 model it is the default and is ignored by :PCG)");
    return BA_ERR_ARG;
  }
  if (o->facto_type < 0 || o->facto_type > 2) {
    ba_set_error("ba_lm_solve: facto_type must be 0 (eltype(x)), 1 (Float32) or 2 (Float16)");
    return BA_ERR_ARG;
  }
  if (o->facto_type == 2 && (o->variant != 1 || o->facto != 0 || p->comm.active())) {
    ba_set_error("ba_lm_solve: facto_type = Float16 exists in lm.jl's :LDL branch only (src/lm.jl:92-95,165-169), one GPU");
    return BA_ERR_ARG;
  }
  if (o->normalize < 0 || o->normalize > 2) {
    ba_set_error("ba_lm_solve: normalize must be 0 (:None), 1 (:J) or 2 (:A)");
    return BA_ERR_ARG;
  }
  if (o->perm < 0 || o->perm > 2) {
    ba_set_error("ba_lm_solve: perm must be 0 (:AMD), 1 (:Metis) or 2 (the caller's camera numbering)");
    return BA_ERR_ARG;
  }
  BA_HIP_CHECK(hipSetDevice(p->device));
  const double t_start = wall();
  BA_CHECK(set_ordering(p, o->perm));  // (lm_ensure inside)
  LMWorkFull *w = static_cast<LMWorkFull *>(p->lm);
  hipStream_t st = p->stream;
  const int V = o->variant;
  // defaults: src/lm.jl:20-26 / src/LevenbergMarquardt.jl:21-26
  const bool xf32 = o->x_f32 != 0;  // eltype(x) = Float32: eps(T)-derived defaults, Float32 iterates and evaluations
  if (xf32) BA_CHECK(ensure_xf32(p, w));
  w->f16 = V && o->facto_type == 2;
  if (w->f16) BA_CHECK(ensure_f16(p, w));
  const bool facto_f32 = V && o->facto_type >= 1 && o->facto != 2;  // Float16 inputs are eliminated and factored in Float32
  w->pcg = o->facto == 2;
  w->pcg_tol = o->pcg_tol > 0 ? o->pcg_tol : 1e-8;
  w->pcg_maxit = o->pcg_max_iter > 0 ? o->pcg_max_iter : 0;
  w->n_cg = 0;
  if (!w->pcg) BA_CHECK(ensure_dense(p, w));  // (here, not in linear_step: no allocation while a graph is being recorded)
  // Scalars carry the width Julia's promotion rules give them (TS: value + 32 | 64).  For a Float64 model everything is
  // a Float64 and the arithmetic below is plain double arithmetic.  For eltype(x) = Float32 (src/lm.jl:20-26,36-59):
  // norm() of a Float32 vector, obj = norm_r^2 / 2, pred, ared, rho are Float32; the eps(Float32)-derived default
  // tolerances and nu_d, nu_m, delta_d, lambda are Float32 VALUES whose expressions run in Float32, while a value the
  // caller passes is a Float64 (the reference's own Float32 experiment passes Float64 literals, src/diffprecsions.jl:22) and
  // promotes its expression; lambda = T(max(lambda, 1e10 / norm_Jtr)) (lm.jl:59) stays a Float32 until the first accepted
  // step, whose `max(1.0e-8, lambda)` (lm.jl:337, a Float64 literal) makes it a Float64 for the rest of the run; in
  // LevenbergMarquardt.jl only Float32 operations touch it.  The accept tests multiply by Float64 literals (lm.jl:259,335).
  using ts::TS;
  const int W = xf32 ? 32 : 64;
  auto T_ = [&](double v) { return xf32 ? ts::f32(v) : ts::f64(v); };  // a value of type eltype(x)
  const TS eps = T_(xf32 ? 1.1920928955078125e-07 : 2.220446049250313e-16), sq = ts::sqrt(eps);
  const TS cbr = T_(std::pow(eps.v, 1.0 / 3.0));  // eltype(x)(eps^(1/3)): a Float64 power, converted
  const TS c100 = ts::mul(T_(100), sq), c1000 = ts::mul(T_(1000), sq);
  auto opt = [&](double given, TS dflt, bool positive) { return (positive ? given > 0 : given >= 0) ? ts::f64(given) : dflt; };
  const TS restol = opt(o->restol, V ? cbr : c100, false), satol = opt(o->satol, sq, false), srtol = opt(o->srtol, sq, false);
  const TS oatol = opt(o->oatol, sq, false), ortol = opt(o->ortol, V ? cbr : c1000, false);
  const TS atol = opt(o->atol, V ? sq : c100, false), rtol = opt(o->rtol, V ? cbr : c1000, false);
  const TS nu_d = opt(o->nu_d, T_(3), true), nu_m = opt(o->nu_m, T_(3), true), delta_d = opt(o->delta_d, T_(2), true);
  TS lambda = opt(o->lambda, T_(V ? 30 : 0.1), true);
  const int ite_max = o->ite_max >= 0 ? o->ite_max : (V ? 200 : 100);
  const bool linesearch = V && o->linesearch;
  const bool facto_qr = o->facto >= 1;  // same device solve; the branches differ in the line search's model value only
                                        // (:PCG has no dr vector to recur on either: it re-evaluates like :QR)
  // norm(v) from the device's sum of squares; 1/2 |v|^2 the way the reference forms it (norm(v)^2 / 2: for a Float32
  // vector the norm is rounded to Float32 first; for Float64 the sum of squares is halved directly, as before)
  auto norm_of = [&](double sumsq, int wd) { return wd == 32 ? ts::f32(std::sqrt(sumsq)) : ts::f64(std::sqrt(sumsq)); };
  auto half_of = [&](double sumsq) {
    if (!xf32) return ts::f64(0.5 * sumsq);
    const TS n = ts::f32(std::sqrt(sumsq));
    return ts::div(ts::mul(n, n), ts::f32(2));
  };

  memset(stats, 0, sizeof *stats);
  stats->status = BA_ST_UNKNOWN;
  double *h_sh = w->s.h_sh, *h_rp = w->s.h_rp;

  BA_HIP_CHECK(hipMemcpyAsync(w->x, x_inout, (size_t)w->nvar * sizeof(double), x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  if (xf32) {  // x0 is a Float32 vector at the reference's boundary; make sure of it
    BA_CHECK(launch_convert(w->x, w->xf, w->nvar, st));
    BA_CHECK(launch_convert(w->xf, w->x, w->nvar, st));
  }
  BA_CHECK(refresh_linearisation(p, w, true, st, xf32));  // r, J, J'r   (lm.jl:39-58)
  BA_CHECK(fetch_scalars(p, w, st));
  stats->n_residual++;
  stats->n_jacobian++;
  TS norm_r = norm_of(h_sh[SH_RSQ], W);
  TS obj = ts::div(ts::mul(norm_r, norm_r), T_(2));
  TS norm_Jtr = norm_of(h_sh[SH_GP] + h_rp[RP_GC], W);
  TS norm_x = norm_of(h_sh[SH_X_P] + h_rp[RP_X_C], W);
  if (V) {  // lm.jl:59: lambda = T(max(lambda, 1e10 / norm_Jtr))
    lambda = ts::max(lambda, ts::div(ts::f64(1e10), norm_Jtr));
    lambda = T_(lambda.v);
  }

  TS norm_delta = T_(0), dr2 = T_(0), ared = T_(0), pred = T_(0);
  const TS eps_first = ts::add(atol, ts::mul(rtol, norm_Jtr));  // lm.jl:107
  TS old_obj = obj;
  bool small_step = false, first_order = norm_Jtr.v < eps_first.v, small_residual = norm_r.v < restol.v;
  bool small_obj_change = false, fail2 = false;
  int iter = 0;
  bool tired = iter > ite_max;
  bool accepted = false, have_trial = false;
  int rc = BA_OK;
  const double t_loop = wall();

  while (!(small_step || first_order || small_residual || small_obj_change || tired || fail2)) {
    if (V) iter++;                                                                           // lm.jl:127
    if (!V && cb) cb(cb_ctx, iter, obj.v, ts::sub(old_obj, obj).v, norm_Jtr.v, lambda.v, norm_delta.v, dr2.v, accepted);  // LevenbergMarquardt.jl:143-147
    if (have_trial) have_trial = false;  // submitted with the refresh of the step accepted last (accept_refresh_and_trial)
    else if ((rc = trial_step(p, w, lambda.v, o->normalize, facto_f32, xf32, st)) != BA_OK) break;  // lm.jl:154-254
    stats->n_factor++;
    stats->n_residual++;
    if (*w->h_flag == 2) {
      ba_set_error("dense factorisation: a hoisted diagonal tile never became ready (internal scheduling error)");
      rc = BA_ERR_HIP;
      break;
    }
    if (*w->h_flag) {
      ba_set_error("reduced camera system: exactly zero pivot (SQDException in the reference)");
      rc = BA_ERR_ZERO_PIVOT;
      break;
    }
    dr2 = half_of(h_sh[SH_MODEL]);  // 1/2 |delta_r|^2   (lm.jl:229)
    TS obj_suiv = half_of(h_sh[SH_RSQ_TRIAL]);
    TS norm_rsuiv = norm_of(h_sh[SH_RSQ_TRIAL], W);
    if (!V) iter++;  // LevenbergMarquardt.jl:240
    // the reference's delta is a Vector{Float32} for a Float32 model -- except under normalize = :A once lambda is a
    // Float64 (delta /= sqrt(lambda), lm.jl:235-237) or behind a Float64 delta_d in the line search (lm.jl:266)
    int delta_w = (xf32 && !(V && o->normalize == 2 && lambda.w == 64)) ? 32 : 64;

    bool step_accepted;
    int ntimes = 0;
    if (V) {
      pred = ts::sub(obj, dr2);
      ared = ts::sub(obj, obj_suiv);
      step_accepted = ared.v >= 1e-4 * pred.v;  // lm.jl:257-259 (Float64 literal)
      double c_r = w->cr0();  // delta_r = -(J delta + c_r r)   (1; 1/mu in the Float16 branch)
      while (linesearch && !step_accepted && ntimes < 4) {  // lm.jl:264-295
        // delta /= delta_d ; delta_r = (delta_r - r)/delta_d (lm.jl:277): with delta_r = -(J delta + c r) the update is
        // c <- (c + 1)/delta_d, which stays 1 only for the default delta_d = 2 (the reference's comment at lm.jl:275-276
        // assumes it; the code is followed, not the comment)
        // The :QR branch recomputes |J delta + r|^2 instead (lm.jl:273): c stays 1 there.
        if (!facto_qr) c_r = (c_r + 1.0) / delta_d.v;
        if (delta_d.w == 64) delta_w = 64;
        if ((rc = launch_scale_scalar(p, w->nvar, w->delta, 1.0 / delta_d.v, st)) != BA_OK) break;
        if ((rc = step_scalars(p, w, st, c_r)) != BA_OK) break;
        if ((rc = trial_point(p, w, st, xf32)) != BA_OK) break;
        stats->n_residual++;
        if ((rc = comm_sum(p, w, w->s.off_scal + SH_TRIAL_FIRST, SH_TRIAL_COUNT, st)) != BA_OK) break;
        if ((rc = fetch_scalars(p, w, st)) != BA_OK) break;
        dr2 = half_of(h_sh[SH_MODEL]);
        obj_suiv = half_of(h_sh[SH_RSQ_TRIAL]);
        norm_rsuiv = norm_of(h_sh[SH_RSQ_TRIAL], W);
        pred = ts::sub(obj, dr2);
        ared = ts::sub(obj, obj_suiv);
        step_accepted = ared.v >= 1e-4 * pred.v;
        ntimes++;
      }
      if (rc != BA_OK) break;
    } else {
      step_accepted = ts::sub(obj_suiv, obj).v < 1e-4 * ts::sub(dr2, obj).v;  // LevenbergMarquardt.jl:243
    }
    accepted = step_accepted;
    const TS nd = norm_of(h_sh[SH_DELTA_P] + h_rp[RP_DELTA_C], delta_w);

    if (V) {
      norm_delta = nd;  // lm.jl:297-302
      if (std::isnan(norm_delta.v)) {
        fail2 = true;
        continue;
      }
      const double rho = ts::div(ared, pred).v;
      if (cb) cb(cb_ctx, iter, obj.v, ts::sub(old_obj, obj).v, norm_Jtr.v, lambda.v, norm_delta.v, rho,
                 (step_accepted && dr2.v <= obj.v) ? 1 : 0);  // lm.jl:304
      if (o->verbose)
        fprintf(stderr, "%6d %14.7e %10.2e %10.2e %10.2e %10.2e %10.2e %s\n", iter, obj.v, ts::sub(old_obj, obj).v, norm_Jtr.v,
                lambda.v, norm_delta.v, rho, (step_accepted && dr2.v <= obj.v) ? "acc" : "rej");
    }

    if (!step_accepted) {
      stats->n_rejected++;
      if (V) lambda = ts::mul(ts::max(lambda, ts::div(norm_delta.w == 32 ? ts::f32(1) : ts::f64(1), norm_delta)), ts::powi(nu_m, ntimes + 1));  // lm.jl:308
      else lambda = ts::mul(lambda, nu_m);                                                            // LevenbergMarquardt.jl:269
    } else {
      stats->n_accepted++;
      if (V) {  // lm.jl:329-337
        if (ntimes > 0) lambda = ts::div(lambda, ts::powi(nu_d, ntimes - 1));
        else lambda = ts::div(lambda, nu_d);
        if (ared.v >= 0.9 * pred.v) lambda = ts::div(lambda, nu_d);
        lambda = ts::max(ts::f64(1.0e-8), lambda);  // the Float64 literal promotes lambda for the rest of the run
      } else {
        lambda = ts::div(lambda, nu_d);  // LevenbergMarquardt.jl:292
      }
      std::swap(w->x, w->x_trial);  // x .= x_suiv
      std::swap(w->r, w->r_trial);  // r .= r_suiv
      w->parity ^= 1;
      old_obj = obj;
      norm_r = norm_rsuiv;
      obj = obj_suiv;
      // (what of the stopping tests is known before the refresh decides whether the next trial step goes out with it)
      const bool stop_known = norm_r.v < restol.v || ts::sub(old_obj, obj).v < ts::add(oatol, ts::mul(ortol, old_obj)).v || iter > ite_max;
      if (!stop_known && can_prefetch_trial(p, w, o->normalize, facto_f32, xf32)) {
        if ((rc = accept_refresh_and_trial(w, lambda.v, facto_f32, st)) != BA_OK) break;
        have_trial = true;
      } else if ((rc = accept_refresh(p, w, xf32, st)) != BA_OK) break;  // J, J'r  (lm.jl:341,370)
      stats->n_jacobian++;
      norm_Jtr = norm_of(h_sh[SH_GP] + h_rp[RP_GC], W);
      norm_x = norm_of(h_sh[SH_X_P] + h_rp[RP_X_C], W);
      if (!V) norm_delta = nd;  // LevenbergMarquardt.jl:352
      small_step = norm_delta.v < ts::add(satol, ts::mul(srtol, norm_x)).v;  // lm.jl:375-379
      first_order = norm_Jtr.v < eps_first.v;
      small_residual = norm_r.v < restol.v;
      small_obj_change = ts::sub(old_obj, obj).v < ts::add(oatol, ts::mul(ortol, old_obj)).v;
    }
    if (!V && o->verbose)
      fprintf(stderr, "%6d %14.7e %10.2e %10.2e %10.2e %10.2e %10.2e %s\n", iter, obj.v, ts::sub(old_obj, obj).v, norm_Jtr.v,
              lambda.v, nd.v, dr2.v, step_accepted ? "true" : "false");
    tired = iter > ite_max;  // lm.jl:382
  }
  if (rc == BA_OK && !V && cb) cb(cb_ctx, iter, obj.v, ts::sub(old_obj, obj).v, norm_Jtr.v, lambda.v, norm_delta.v, dr2.v, accepted);
  stats->loop_s = wall() - t_loop;

  if (rc != BA_OK) {
    stats->status = BA_ST_EXCEPTION;
  } else if (small_step) stats->status = BA_ST_SMALL_STEP;  // lm.jl:391-405
  else if (first_order) stats->status = BA_ST_FIRST_ORDER;
  else if (small_residual) stats->status = BA_ST_SMALL_RESIDUAL;
  else if (small_obj_change) stats->status = BA_ST_ACCEPTABLE;
  else if (fail2) stats->status = BA_ST_EXCEPTION;
  else if (tired) stats->status = BA_ST_MAX_ITER;
  stats->iter = iter;
  stats->objective = obj.v;
  stats->dual_feas = norm_Jtr.v;
  stats->lambda_final = lambda.v;
  stats->n_cg = (int)w->n_cg;
  w->pcg = false;
  {
    hipError_t e = hipMemcpyAsync(x_inout, w->x, (size_t)w->nvar * sizeof(double), x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess && rc == BA_OK) {
      ba_set_error("solution copy: %s", hipGetErrorString(e));
      rc = BA_ERR_HIP;
    }
  }
  stats->elapsed_s = wall() - t_start;
  return rc;  // a NaN step is not an error of the call: status :exception, as in the reference (lm.jl:297-302,401-402)
}
