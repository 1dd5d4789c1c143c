// Internal: LM workspace and launchers of the normal-equation kernels.
#pragma once
#include "ba_internal.h"

constexpr int RED_BLOCKS = 1024;  // fixed number of partial sums => fixed summation tree

// Schur task list (built once per problem on the host): all ordered observation pairs (a, b) of one point with
// camera(a) >= camera(b), sorted by (camera(a), camera(b)); key k owns tasks [key_ptr[k], key_ptr[k+1]).
// Every camera has its diagonal key even when it has no observation.
// Long keys (few cameras, many shared points: Dubrovnik-356 has ~80 tasks per key, LadyBug-49 ~70) are split into chunks of
// `chunk` tasks, one wave per chunk, whose partial 9x9 sums are added up in chunk order by a second kernel: one wave per
// key would leave most of the chip idle there (63 k waves of 40 dependent iterations each).  Keys of at most 2 chunks' worth
// of tasks are summed by one wave as before.
struct SchurTasks {
  int64_t nkeys = 0, ntasks = 0;
  int *key_ptr = nullptr, *key_ca = nullptr, *key_cb = nullptr;  // device
  int *task_a = nullptr, *task_b = nullptr;                       // device
  int chunk = 0;                                                  // tasks per chunk; keys with more than 2 * chunk tasks are split
  int64_t nsplit = 0, nchunks = 0;
  int *skey = nullptr, *skey_c0 = nullptr;                        // split keys: key id, first chunk (nsplit + 1 entries)
  int *chunk_t0 = nullptr, *chunk_t1 = nullptr;                   // chunks: task range
  double *partial = nullptr;                                      // nchunks x 81
  std::vector<unsigned char> tile_occ;                            // host: nt x nt lower tile occupancy of S by the keys (before fill)
  // Fill-reducing camera ordering (ba_order.cpp): key_ca / key_cb are BLOCK ROWS of S; cam_of[k] (device; null = identity)
  // is the camera at block row k, pos its inverse.  Only the reduced camera system lives in that order: S, its right-hand
  // side and solution, the column scaling; x, J, Hcc, gc stay in the caller's camera order.
  int *cam_of = nullptr, *pos = nullptr;
  std::vector<int> h_key_cb, h_skey;                              // host copies (chunked assembly of a distributed run)
};

// one chunk of tile columns of the reduced camera matrix in a distributed run with per-rank ownership of S: the columns
// [owner's local tile range t0 .. t0 + ntiles), the keys whose 9 x 9 blocks touch them, its offset table (negative: not here)
struct SchurChunk {
  int owner = 0;           // -1: a reduce-scatter chunk -- one segment per owner, `seg` tiles each, laid out by rank
  int64_t t0 = 0, ntiles = 0;
  int64_t seg = 0, my_t0 = 0, my_n = 0;  // reduce-scatter chunk: tiles per segment; where this rank's segment goes in its own S
  int64_t *cco = nullptr;  // device, nt entries (cco_alloc + 1: tix reads the table's head one entry before)
  int64_t *cco_alloc = nullptr;
  int *keys = nullptr, *skeys = nullptr;  // device: key ids / indices into the split-key list
  int64_t nkeys = 0, nskeys = 0;
};

// up to six sums of squares in one launch pair (launch_sumsq_multi): vector, length, destination array and slot; the
// second kernel (one workgroup) can also publish the controller's scalars to pinned host memory once every sum is in place
constexpr int SUMSQ_JOBS = 6;
struct SumsqJobs {
  int count = 0;
  const double *v[SUMSQ_JOBS] = {};
  int64_t n[SUMSQ_JOBS] = {};
  double *out[SUMSQ_JOBS] = {};
  int slot[SUMSQ_JOBS] = {};
  int nb[SUMSQ_JOBS] = {};  // filled by the launcher
  // publish (optional; as launch_publish): a[0..na) -> ha, b[0..nb2) -> hb, flag[0] -> hflag
  const double *pa = nullptr, *pb = nullptr;
  double *ha = nullptr, *hb = nullptr;
  const int *pflag = nullptr;
  int *hflag = nullptr;
  int na = 0, nb2 = 0;
  void add(const double *vec, int64_t len, double *dst, int s) {
    v[count] = vec;
    n[count] = len;
    out[count] = dst;
    slot[count] = s;
    count++;
  }
  void publish(const double *a, int na_, double *h_a, const double *b, int nb_, double *h_b, const int *flag, int *h_flag) {
    pa = a; na = na_; ha = h_a; pb = b; nb2 = nb_; hb = h_b; pflag = flag; hflag = h_flag;
  }
};

// scalar slots of LMWork::scal (device) / h_scal (pinned host)
enum { SC_RSQ = 0, SC_RSQ_TRIAL, SC_MODEL, SC_DELTA, SC_XSQ, SC_JTR, SC_COUNT = 8 };

struct LMWork {
  int64_t nvar = 0, nequ = 0, n = 0, npad = 0;  // n = 9*ncams
  double *x = nullptr, *x_trial = nullptr, *delta = nullptr;
  double *r = nullptr, *r_trial = nullptr, *J = nullptr;
  double *Hpp = nullptr, *gp = nullptr, *Uinv = nullptr, *u = nullptr;
  double *Yobs = nullptr;                // 6/obs: U^-1 A_b' of the current damping
  bool model_done = false;               // the step's model value was formed by the back-substitution pass
  double *Hcc = nullptr, *gc = nullptr;  // gc: 9*ncams
  double *hdiag = nullptr;               // npad: diag of the camera block of J'J summed over all ranks (column scalings)
  double *rhs = nullptr;                 // npad
  double *colscale = nullptr;            // nvar (normalize != None)
  // facto_type = Float16: |J_j|^2, column norms, damping vector (nvar each), quantised J (24/obs) and r; allocated on first use
  double *jn2 = nullptr, *dcol = nullptr, *damp = nullptr, *Jq = nullptr, *rq = nullptr;
  double *partial = nullptr;             // RED_BLOCKS
  double *partial_multi = nullptr;       // SUMSQ_JOBS x RED_BLOCKS (launch_sumsq_multi)
  int *cam_pnt = nullptr;                // nobs: the point of every observation in camera order (pnt0[cam_obs[q]])
  double *scal = nullptr;                // SC_COUNT device scalars
  double *h_scal = nullptr;              // pinned host mirror
  SchurTasks tasks;
  DenseLDL ldl;
};

// d_lambda (optional device scalar): the damping used is lambda * d_lambda[0] (hipGraph replays, ba_lm.hip)
int launch_schur_prep(ba_problem *p, double lambda, const double *d_Hpp, const double *d_gp, double *d_Uinv,
                      double *d_u, hipStream_t st, const double *d_lambda = nullptr, const double *d_damp = nullptr,
                      double *d_lambda_copy = nullptr /* d_lambda (may be pinned host memory) is copied here */,
                      double *d_zero = nullptr, int64_t nzero = 0 /* cleared on the way */);
int launch_schur_blocks(ba_problem *p, const SchurTasks *T, const double *d_J, const double *d_Uinv, double *d_Y,
                        const double *d_Hcc, double lambda, double *d_S, const int64_t *d_col_off, int64_t n, int64_t npad,
                        hipStream_t st, const double *d_lambda = nullptr, const double *d_damp = nullptr, int64_t s_tiles = 0);
// facto_type = Float16 (ba_normal_kernels.hip, k_f16_cols): |J_j|^2 of every column; column scaling + Float16 rounding
int launch_col_sq(ba_problem *p, const double *d_Hpp, const double *d_hdiag, double *d_jn2, hipStream_t st);
int launch_f16_scale(ba_problem *p, double lambda, double mu, const double *d_jn2, const double *d_J, const double *d_r,
                     double *d_dcol, double *d_damp, double *d_Jq, double *d_rq, hipStream_t st);
int launch_schur_pre(ba_problem *p, const SchurTasks *T, const double *d_J, const double *d_Uinv, double *d_Y, hipStream_t st);
int launch_schur_chunk(ba_problem *p, const SchurTasks *T, const SchurChunk *c, const double *d_J, const double *d_Y,
                       const double *d_Hcc, double lambda, double *dest, int64_t n, int64_t npad, hipStream_t st,
                       const double *d_lambda = nullptr, const double *d_damp = nullptr);
int launch_scale_S_own(ba_problem *p, int64_t n, const double *d_dsc, double *d_S, const int64_t *d_col_off, const int *d_own_cols,
                       const int64_t *d_own_pref, int ncols, int64_t ntiles, hipStream_t st);
int launch_scale_S_list(ba_problem *p, int64_t n, const double *d_dsc, double *d_S, const int2 *d_tiles, int64_t ntiles, hipStream_t st);
int launch_schur_rhs(ba_problem *p, const double *d_J, const double *d_r, const double *d_u, double *d_rhs,
                     hipStream_t st, const int *d_cam_pnt = nullptr, const int *d_pos = nullptr /* block row of a camera */);
int launch_gather_cams(ba_problem *p, const int *d_pos, const double *d_src, double *d_dst, hipStream_t st);
int launch_backsub(ba_problem *p, const double *d_J, const double *d_Uinv, const double *d_u, const double *d_dc,
                   double *d_dp, hipStream_t st, const double *d_r_model = nullptr, double cr = 1.0, double *d_partial = nullptr,
                   double *d_scal = nullptr, int slot = 0, bool *model_done = nullptr);
int launch_model_sq(ba_problem *p, const double *d_J, const double *d_r, const double *d_delta, double *d_partial,
                    double *d_scal, int slot, hipStream_t st, double cr = 1.0);
int launch_sumsq(ba_problem *p, int64_t n, const double *d_v, double *d_partial, double *d_scal, int slot,
                 hipStream_t st);
// d_a[0..na), d_b[0..nb), d_flag[0] (optional) -> pinned host buffers, one launch
int launch_publish(const double *d_a, int na, double *h_a, const double *d_b, int nb, double *h_b, const int *d_flag, int *h_flag,
                   hipStream_t st);
int launch_sumsq_multi(ba_problem *p, SumsqJobs *jobs, double *d_partial_multi /* SUMSQ_JOBS RED_BLOCKS */, hipStream_t st);
int launch_axpy(ba_problem *p, int64_t n, const double *d_x, const double *d_d, double *d_y, hipStream_t st);
int launch_hcc_diag(ba_problem *p, const double *d_Hcc, double *d_hdiag, hipStream_t st);
int launch_cam_scale(ba_problem *p, const double *d_hdiag, double add, double *d_dsc, hipStream_t st,
                     const double *d_lambda = nullptr, const int *d_pos = nullptr);
int launch_scale_S(ba_problem *p, int64_t n, int64_t nt, const double *d_dsc, double *d_S, const int64_t *d_col_off,
                   hipStream_t st);
int launch_scale_scalar(ba_problem *p, int64_t n, double *d_v, double alpha, hipStream_t st);
int launch_scale_vec(ba_problem *p, int64_t n, const double *d_s, double *d_v, int divide, hipStream_t st);

// preconditioned conjugate gradients on the reduced camera system (facto = PCG; ba_normal_kernels.hip)
int launch_wuw(ba_problem *p, const double *d_J, const double *d_h, const double *d_Hcc, const double *d_v, double lam, double *d_q,
               hipStream_t st, const int *d_cam_pnt = nullptr);
int launch_cam_pnt(ba_problem *p, int *d_cam_pnt, hipStream_t st);
int launch_schur_diag(ba_problem *p, const double *d_J, const double *d_Uinv, const double *d_Hcc, double *d_blk45, hipStream_t st);
int launch_pcg_factor(ba_problem *p, double lambda, double *d_blk45, int *d_flag, hipStream_t st);
int launch_axpy_s(ba_problem *p, int64_t n, double a, const double *d_x, double *d_y, hipStream_t st);
int launch_cg_alpha(ba_problem *p, int64_t n, const double *d_p, const double *d_q, double *d_cg, hipStream_t st);
int launch_cg_step(ba_problem *p, const double *d_cg, const double *d_L45, const double *d_p, const double *d_q, double *d_x,
                   double *d_r, double *d_z, double *d_partial, int first, hipStream_t st);
int launch_cg_beta_dir(ba_problem *p, int64_t n, const double *d_partial, double *d_cg, const double *d_z, double *d_p, int first,
                       hipStream_t st);
int launch_wtv(ba_problem *p, const double *d_J, const double *d_Uinv, const double *d_v, double *d_h, hipStream_t st);
