/*
 * ba_hip.h -- C ABI of libba_hip.so, the MI355X (gfx950) implementation of the hot path of
 * CelestineAngla/BundleAdjustment.jl.  Plain C types only; every entry returns an int status
 * (BA_OK == 0) and never throws.  ba_last_error() returns the text of the last failure of the
 * calling thread.
 *
 * Each entry names the reference interface it replaces (paths relative to the reference root).
 * Conventions are the reference's own at this boundary:
 *   - indices are 1-based int64 (Julia Int);
 *   - the unknown vector is x = [X_1(3) ... X_npnts(3) ; C_1(9) ... C_ncams(9)], camera
 *     C = (r1,r2,r3,t1,t2,t3,k1,k2,f)                      (src/ReadFiles.jl:29-40);
 *   - pt2d and residuals are interleaved (x,y) per observation;
 *   - Jacobian COO entries: 24 per observation, row-major 2x12 block, column order
 *     [X(3), r(3), t(3), k1, k2, f]                        (src/BALNLPModels.jl:137-153,201).
 * Host-pointer entries copy in/out and keep no caller pointer after returning (the Julia GC may
 * move or free the arrays).  *_dev entries take device pointers (hipMalloc'ed by the caller, e.g.
 * a torch tensor's data_ptr, or by ba_dev_malloc) and enqueue on the given hipStream_t
 * (void* 0 = the handle's own stream) without synchronising.
 */
#ifndef BA_HIP_H
#define BA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  BA_OK = 0,
  BA_ERR_ARG = 1,       /* bad argument (null pointer, index out of range, size mismatch) */
  BA_ERR_HIP = 2,       /* HIP runtime error (no device, out of memory, launch failure) */
  BA_ERR_IO = 3,        /* BAL file missing / malformed / bzip2 runtime missing */
  BA_ERR_ZERO_PIVOT = 4,/* LDL^T met an exactly zero pivot: SQDException, src/ldl_aux.jl:45-47,199 */
  BA_ERR_NAN_STEP = 5,  /* |delta| is NaN: status :exception, src/lm.jl:297-302 */
  BA_ERR_COMM = 6       /* RCCL or the caller's communication hook failed */
};

/* status of a Levenberg-Marquardt run: src/lm.jl:391-405, src/LevenbergMarquardt.jl:370-380 */
enum {
  BA_ST_UNKNOWN = -1,
  BA_ST_SMALL_STEP = 0,
  BA_ST_FIRST_ORDER = 1,
  BA_ST_SMALL_RESIDUAL = 2,
  BA_ST_ACCEPTABLE = 3,
  BA_ST_NEG_PRED = 4,
  BA_ST_EXCEPTION = 5,
  BA_ST_MAX_ITER = 6
};

typedef struct ba_problem ba_problem; /* opaque: device mirrors of one BALNLPModel (src/BALNLPModels.jl:79-88) */

const char *ba_last_error(void);
int ba_device_count(int *n);
/* name/CU count of device `dev` (for logs) */
int ba_device_info(int dev, char *name, size_t name_cap, int *n_cu, size_t *hbm_bytes);

/* ---- BAL reader: readfile(filename, T)  src/ReadFiles.jl:9-53 ------------------------------
 * `path` is a full path to problem-*.txt or problem-*.txt.bz2 (the reference prepends
 * <repo>/Data/, ReadFiles.jl:10; the host shim does that).  Two calls: header, then body into
 * caller-allocated arrays of the sizes the header gave.  Indices come back 1-based, cameras
 * re-ordered from the file's (r,t,f,k1,k2) to (r,t,k1,k2,f) exactly as ReadFiles.jl:32-43.
 * The _f32 twin parses every decimal straight to float (parse(Float32, .), single rounding). */
int ba_read_bal_header(const char *path, int64_t *ncams, int64_t *npnts, int64_t *nobs);
int ba_read_bal(const char *path, int64_t ncams, int64_t npnts, int64_t nobs, int64_t *cam_idx1,
                int64_t *pnt_idx1, double *pt2d, double *x0);
int ba_read_bal_f32(const char *path, int64_t ncams, int64_t npnts, int64_t nobs, int64_t *cam_idx1,
                    int64_t *pnt_idx1, float *pt2d, float *x0);

/* ---- model: BALNLPModel(...)  src/BALNLPModels.jl:91-106 ------------------------------------
 * Copies the index arrays and pt2d to device `device` (int32 0-based mirrors + the point- and
 * camera-sorted observation lists used by the deterministic reductions).  In a multi-GPU run every
 * rank creates its own shard: local observations, local points renumbered 1..npnts_local, ALL
 * cameras (cam_idx1 stays global). */
int ba_problem_create(int device, int64_t ncams, int64_t npnts, int64_t nobs, const int64_t *cam_idx1,
                      const int64_t *pnt_idx1, const double *pt2d, ba_problem **out);
void ba_problem_destroy(ba_problem *p);
int ba_problem_dims(const ba_problem *p, int64_t *ncams, int64_t *npnts, int64_t *nobs, int64_t *nvar,
                    int64_t *nequ, int64_t *nnzj);

/* cons!(nlp, x, cx) == residual!(FeasibilityResidual(nlp), x, r)  src/BALNLPModels.jl:115-122,39-55 */
int ba_residual(ba_problem *p, const double *x, double *r);
int ba_residual_f32(ba_problem *p, const float *x, float *r); /* BALNLPModel(file, Float32) */
/* jac_structure!(nlp, rows, cols)  src/BALNLPModels.jl:125-158 : 24*nobs 1-based int64 each */
int ba_jac_structure(ba_problem *p, int64_t *rows, int64_t *cols);
/* jac_coord!(nlp, x, vals)  src/BALNLPModels.jl:161-206 + src/JacobianByHand.jl:5-101, NaN -> 0 */
int ba_jac_coord(ba_problem *p, const double *x, double *vals);
int ba_jac_coord_f32(ba_problem *p, const float *x, float *vals);
/* J' r from COO values: mul_sparse(cols, rows, vals, r, nnzj, nvar)  src/lma_aux.jl:194-212 as
 * called at src/lm.jl:57,370.  jtr has nvar entries in the layout of x. */
int ba_jtr(ba_problem *p, const double *vals, const double *r, double *jtr);

/* device-resident twins (no host copies, no synchronisation): what bench.py times.  The kernels move 16 bytes per
 * lane: arrays aligned to 16 bytes (any hipMalloc / torch allocation is) run at the quoted rates. */
int ba_residual_dev(ba_problem *p, const double *d_x, double *d_r, void *stream);
int ba_residual_f32_dev(ba_problem *p, const float *d_x, float *d_r, void *stream);
int ba_jac_structure_dev(ba_problem *p, int64_t *d_rows, int64_t *d_cols, void *stream);
int ba_jac_coord_dev(ba_problem *p, const double *d_x, double *d_vals, void *stream);
int ba_jac_coord_f32_dev(ba_problem *p, const float *d_x, float *d_vals, void *stream);
int ba_jtr_dev(ba_problem *p, const double *d_vals, const double *d_r, double *d_jtr, void *stream);

/* small device-memory helpers so that a C / Julia caller needs no HIP binding of its own */
int ba_dev_malloc(ba_problem *p, size_t bytes, void **d_ptr);
int ba_dev_free(ba_problem *p, void *d_ptr);
/* Host <-> device copies, ordered on a stream and complete on return: enqueued on the handle's own stream (the _on forms:
 * on `stream`, a hipStream_t; NULL = the handle's stream) behind the work already enqueued there; the call returns when
 * that stream has drained.  A communication hook (ba_lm_set_comm_hook) that stages through host memory must use the _on
 * forms with the stream it was given: the library's streams are non-blocking, so a copy on any other stream is neither
 * ordered behind the producer of the data nor visible to the kernels the library launches next on `stream`. */
int ba_memcpy_h2d(ba_problem *p, void *d_dst, const void *h_src, size_t bytes);
int ba_memcpy_d2h(ba_problem *p, void *h_dst, const void *d_src, size_t bytes);
int ba_memcpy_h2d_on(ba_problem *p, void *stream, void *d_dst, const void *h_src, size_t bytes);
int ba_memcpy_d2h_on(ba_problem *p, void *stream, void *h_dst, const void *d_src, size_t bytes);
int ba_synchronize(ba_problem *p);

/* ---- Levenberg_Marquardt(model, facto, perm, normalize[, linesearch]; kwargs...) --------------
 * variant 0: src/LevenbergMarquardt.jl:16-385 (what solve_ba.jl runs); variant 1: src/lm.jl:15-418.
 * The linear step (J'J + lambda I) delta = -J' r -- which the reference obtains from a sparse LDL^T
 * (or QR) of the augmented system, src/lm.jl:154-238 -- is solved on the device through the
 * point-eliminated (Schur) reduced camera system and a dense blocked LDL^T on the f64 matrix cores;
 * `facto` and `perm` therefore only select behaviour that survives that change: both :QR and :LDL
 * give the same step; `perm` orders what is left to order, the cameras of the reduced system (its
 * residual rows and points are eliminated first, as AMD orders them).  Negative tolerances / zero
 * parameters mean "the variant's default" (eps-derived, src/lm.jl:20-26 and
 * src/LevenbergMarquardt.jl:21-26). */
typedef struct ba_lm_opts {
  int variant;    /* 0 LevenbergMarquardt.jl, 1 lm.jl */
  int facto;      /* 0 :LDL, 1 :QR -- both are served by the same device solve of (J'J + lambda I) delta = -J'r; what
                   *    survives of the branch: inside the line search :QR re-evaluates |J delta + r|^2 (src/lm.jl:273)
                   *    while :LDL uses the recursion of src/lm.jl:277 (they differ for delta_d != 2).
                   *    2 :PCG (an extension, no counterpart in the reference: SURVEY 8f) -- the reduced camera system is
                   *    never formed; block-Jacobi preconditioned conjugate gradients apply it through J (two sweeps per
                   *    iteration) to |residual| <= pcg_tol |right-hand side|: an inexact LM step, judged by the same
                   *    accept test; Float64, no column scaling, model value as :QR.  normalize != :None and an explicit
                   *    facto_type = Float32 on a Float64 model are refused (BA_ERR_ARG), not ignored */
  int normalize;  /* 0 :None, 1 :J, 2 :A  (src/lma_aux.jl:102-178) */
  int linesearch; /* lm.jl only, src/lm.jl:264-295 */
  int facto_type; /* lm.jl only, `facto_type` keyword: 0 = Float64 (the default for a Float64 model), 1 = Float32
                   *    (src/lm.jl:170-173, src/diffprecsions.jl:39-41; the default for a Float32 model), 2 = Float16
                   *    (src/lm.jl:165-169, src/lma_aux.jl:30-95: columns of K scaled by their norms and by mu = 6550,
                   *    entries and right-hand side rounded to Float16, step taken unscaled -- the device rounds the same
                   *    inputs and then eliminates / factors in Float32; :LDL only, one GPU) */
  int ite_max;    /* <0: default (200 / 100) */
  int verbose;    /* 1: print the reference's log columns to stderr */
  int x_f32;      /* 1: eltype(x) = Float32 (a BALNLPModel(file, Float32) run): iterates rounded to Float32, residual and
                   *    Jacobian by the Float32 kernels, eps(Float32)-derived default tolerances; x_inout stays double */
  double restol, satol, srtol, oatol, ortol, atol, rtol; /* <0: default */
  double nu_d, nu_m, lambda, delta_d;                    /* <=0: default (3, 3, 30 | 0.1, 2) */
  double max_time;                                        /* <=0: 3600 (inert in the reference, lm.jl:33,115,382) */
  double pcg_tol;                                         /* facto = 2: relative residual of the CG solve, <=0: 1e-8 */
  int pcg_max_iter;                                       /* facto = 2: CG iterations per LM step, <=0: 1000 */
  int perm;       /* fill-reducing ordering (src/lm.jl:84-88, src/LevenbergMarquardt.jl:106-110): 0 :AMD, 1 :Metis,
                   *    2 the caller's camera numbering.  Applied to the cameras of the reduced camera system
                   *    (ba_schur_ordering); x, J and every vector at this boundary keep the caller's numbering */
} ba_lm_opts;

typedef struct ba_lm_stats {
  int status;  /* BA_ST_* */
  int iter;    /* LM iterations (lm.jl:127 / LevenbergMarquardt.jl:240) */
  int n_accepted, n_rejected;
  int n_residual, n_jacobian, n_factor;
  int n_cg;    /* facto = 2: CG iterations over the whole solve */
  double objective;    /* 1/2 |r|^2 at the returned x */
  double dual_feas;    /* |J' r| (lm.jl:415; primal_feas in the old variant, LevenbergMarquardt.jl:384) */
  double lambda_final;
  double elapsed_s;    /* whole call, host wall clock */
  double loop_s;       /* the while-loop only (what iter / time is quoted on) */
} ba_lm_stats;

/* one log row per iteration, the reference's columns (src/lm.jl:120-121,304):
 * iter, f, delta_f, |J'r|, lambda, |delta|, rho = ared/pred (variant 0: 1/2|dr|^2), accepted(1/0) */
typedef void (*ba_log_cb)(void *ctx, int iter, double f, double df, double norm_jtr, double lambda,
                          double norm_delta, double rho, int accepted);

/* x_inout: nvar doubles, x0 in, solution out (the `x=` keyword of src/lm.jl:20). */
int ba_lm_solve(ba_problem *p, const ba_lm_opts *opts, double *x_inout, ba_lm_stats *stats, ba_log_cb cb,
                void *cb_ctx);
/* the same with the iterate resident on the device (d_x_inout: nvar doubles of device memory, e.g. from ba_dev_malloc): no
 * host <-> device copy of x on either side of the loop.  What bench.py times (inputs resident in HBM when the timed region
 * starts); a host that keeps x in its own memory, as the reference does, calls ba_lm_solve. */
int ba_lm_solve_dev(ba_problem *p, const ba_lm_opts *opts, double *d_x_inout, ba_lm_stats *stats, ba_log_cb cb,
                    void *cb_ctx);

/* ---- multi-GPU: observations sharded by point, cameras replicated --------------------------------
 * One process per GPU.  Every rank creates its own shard (ba_problem_create: local observations and points, ALL
 * cameras) and attaches a communicator BEFORE its first solve.  Per LM iteration the ranks then exchange camera-side
 * data only: short Float64 all-reduces (J'r camera part, diag(J'J), right-hand side, scalars), one reduce per rank of
 * the part of the reduced camera matrix that rank owns, and the broadcast of each factored panel pair from its owner
 * (the dense factorisation is distributed over the tile-column pairs, owner of pair q = q mod world; the triangular
 * solves are replicated).  The reference is single-process: none of this has a counterpart there.
 *   ba_lm_set_comm_rccl : RCCL over xGMI, called directly from the library on its own stream.  Rank 0 obtains the
 *                         128-byte id with ba_comm_get_unique_id and hands it to the other ranks by any means the host
 *                         has (torch.distributed, MPI.jl, a file); the call is collective (ncclCommInitRank).
 *   ba_lm_set_comm_hook : the host carries the data.  op: BA_COMM_ALLREDUCE_F64 (count doubles, sum, in place),
 *                         BA_COMM_REDUCE_F64 / BA_COMM_REDUCE_F32 (count doubles / floats, the sum lands on `root` only,
 *                         in place; the Float32 form carries the reduced camera matrix of facto_type = Float32 runs),
 *                         BA_COMM_BCAST_BYTES (count bytes from `root`), BA_COMM_REDUCE_SCATTER_F64 / _F32 (see the enum).
 *                         d_buf is a device pointer; the operation must
 *                         be ordered after prior work on `stream` and complete (or stream-ordered ON `stream`) on return:
 *                         `stream` is not always the handle's main stream (the distributed factorisation hands its
 *                         panels over on a second, transfer stream while the trailing update runs on the main one), and
 *                         both are hipStreamNonBlocking -- a host-staged hook copies with ba_memcpy_d2h_on /
 *                         ba_memcpy_h2d_on(p, stream, ...), never on the null stream. */
enum { BA_COMM_ALLREDUCE_F64 = 0, BA_COMM_REDUCE_F64 = 1, BA_COMM_BCAST_BYTES = 2, BA_COMM_REDUCE_F32 = 3,
       /* d_buf holds `world` segments of `count` doubles / floats; on return segment `rank` (at d_buf + rank * count)
        * holds the sum of that segment over the ranks, the other segments are unspecified; `root` is unused.  What the
        * assembly of the reduced camera matrix uses: every link of every GPU carries 1 / world of a chunk */
       BA_COMM_REDUCE_SCATTER_F64 = 4, BA_COMM_REDUCE_SCATTER_F32 = 5, BA_COMM_OPS = 6 };
#define BA_COMM_ID_BYTES 128
typedef int (*ba_comm_fn)(void *ctx, int op, void *d_buf, int64_t count, int root, void *stream);
int ba_comm_get_unique_id(void *id_out /* BA_COMM_ID_BYTES */);
int ba_lm_set_comm_rccl(ba_problem *p, int rank, int world, const void *id /* BA_COMM_ID_BYTES */);
int ba_lm_set_comm_hook(ba_problem *p, int rank, int world, ba_comm_fn fn, void *ctx);
/* Layout of the reduced camera matrix S (lower triangle of 128 x 128 tiles, nt = ceil(9 ncams / 128) tile rows) over
 * `world` ranks: tile (i, j), j <= i, sits at tile offset col_off[j] + (i - j); the tile columns are taken in pairs
 * (2q, 2q+1) owned by rank q % world, and rank r's columns fill the contiguous tile range [own_range[r], own_range[r+1]).
 * col_off: nt entries, own_range: world + 1 entries (may be NULL).  Host-only, needs no device. */
int ba_dist_layout(int64_t nt, int world, int64_t *col_off, int64_t *own_range);
/* number of transport calls / bytes handed to the transport since the communicator was attached */
int ba_comm_stats(ba_problem *p, int64_t *calls, int64_t *bytes);
/* the same per operation kind: calls[BA_COMM_OPS], bytes[BA_COMM_OPS], indexed by the BA_COMM_* codes (bytes: what one rank
 * hands to the transport) */
int ba_comm_stats_ops(ba_problem *p, int64_t *calls, int64_t *bytes);

/* ---- single linear step, exposed for parity tests and profiling ------------------------------------
 * From (x, lambda): delta (nvar) solving (J'J + lambda I) delta = -J' r, and pred2 = 1/2 |J delta + r|^2
 * (== 1/2 |delta_r|^2 of the reference's augmented solve, src/lm.jl:229). */
int ba_lm_step(ba_problem *p, const double *x, double lambda, double *delta, double *half_sq_model,
               double *jtr /* nvar or NULL */);
/* the same step with facto_type = Float32 (src/lm.jl:170-173, src/diffprecsions.jl:39-41): the reduced camera system is
 * rounded to Float32, factored and solved there; everything else stays Float64 */
int ba_lm_step_f32(ba_problem *p, const double *x, double lambda, double *delta, double *half_sq_model,
                   double *jtr /* nvar or NULL */);
/* the same step by facto = PCG (see ba_lm_opts.facto): tol / max_iter as pcg_tol / pcg_max_iter; cg_iters_out may be NULL */
int ba_lm_step_pcg(ba_problem *p, const double *x, double lambda, double tol, int max_iter, double *delta,
                   double *half_sq_model, double *jtr, int *cg_iters_out);

/* Block-sparse reduced camera system.  Cameras that share no point leave empty 9 x 9 blocks in S; the reference's sparse
 * LDL^T exploits that (symbolic phase src/ldl_aux.jl:82-119, numeric :122-201).  The device keeps the sparsity at the
 * granularity of its 128 x 128 tiles: on the first direct solve the tile occupancy of the Schur key list goes through a
 * symbolic factorisation per tile column pair, and when the pattern's trailing updates are at most 60 % of the dense
 * factorisation's, assembly-side zeros are skipped by a list-driven schedule and only the pattern's tiles are allocated
 * (one GPU; BA_SPARSE_S=1 / 0 forces it on / off).  tile_fill: pattern tiles (with fill) / all lower tiles; flop_fill: trailing-update tiles of the pattern / of the
 * dense factorisation; sparse_schedule: 1 when the list schedule is in use.  Valid after the first direct solve. */
int ba_lm_schur_pattern(ba_problem *p, double *tile_fill, double *flop_fill, int *sparse_schedule);

/* ---- fill-reducing camera ordering: `perm` = :AMD / :Metis, src/lm.jl:84-88, consumed by ldl_analyse,
 * src/ldl_aux.jl:246-283 ---------------------------------------------------------------------------------
 * The reference orders the augmented matrix K with AMD.jl or Metis.jl (third-party C libraries, absent here).  On the
 * device the residual rows and the points are eliminated first, in closed form; what remains to be ordered is the camera
 * graph (cameras adjacent when they share a point), and an ordering is judged by the 128 x 128 TILE pattern it leaves.
 * method 0 (:AMD): minimum degree with an elimination-tree postorder; 1 (:Metis): nested dissection by level-structure
 * separators; both also offer reverse Cuthill-McKee sequences (hub cameras deferred) and the caller's numbering, and keep
 * the sequence whose symbolic factorisation is cheapest; 2: the caller's numbering.  Only the reduced camera system is
 * permuted (inside the handle); every array at this boundary keeps the caller's camera numbering.
 *   ba_schur_ordering    : host only, no device: the ordering of a problem and the fill it leaves.  perm1 (ncams, may be
 *                          NULL): perm1[k] = the 1-based camera at block row k of S; tile_fill / flop_fill as
 *                          ba_lm_schur_pattern; block_fill: camera pairs sharing a point / all pairs.
 *   ba_lm_set_ordering   : ordering of the handle's next direct solves (default 0; ba_lm_solve sets it from opts.perm).
 *                          Changing it after a direct solve drops the handle's reduced-system workspace.
 *   ba_lm_schur_ordering : the sequence in use after the first direct solve, and the name of the candidate that won
 *                          (static storage). */
int ba_schur_ordering(int64_t ncams, int64_t npnts, int64_t nobs, const int64_t *cam_idx1, const int64_t *pnt_idx1,
                      int method, int64_t *perm1, double *tile_fill, double *flop_fill, double *block_fill);
int ba_lm_set_ordering(ba_problem *p, int method);
int ba_lm_schur_ordering(ba_problem *p, int64_t *perm1, const char **name);
/* What a handle holds of the reduced camera matrix, in 128 x 128 tiles of its scalar type (Float64; a Float32 factorisation
 * adds half of that again): tiles_full = the whole lower triangle, nt (nt + 1) / 2; tiles_held = what this handle allocated
 * for S; tiles_staging = the staging buffer of the chunked assembly.  One GPU (and BA_DIST_FACTOR=0): held = full,
 * staging = 0 -- unless the block-sparse list schedule is in use, which allocates the pattern's tiles only.  Distributed factorisation: a rank holds its own tile columns only (about full / world) plus a staging
 * chunk of at most half of that -- per-rank ownership of S.  Valid after the first direct solve. */
int ba_lm_schur_memory(ba_problem *p, int64_t *tiles_full, int64_t *tiles_held, int64_t *tiles_staging);

/* ---- per-kernel timing (hipEvent pairs on the handle's stream) ------------------------------------ */
int ba_profile_enable(ba_problem *p, int on);
int ba_profile_reset(ba_problem *p);
/* fills up to cap entries; returns the number of kernel classes in *n.  names[i] points into static storage. */
int ba_profile_get(ba_problem *p, int cap, const char **names, double *total_ms, int64_t *calls, int *n);

/* dense blocked LDL^T on the f64 matrix cores, exposed for tests / roofline measurement:
 * factor the symmetric n x n matrix whose lower triangle is given row-major (ld = n) in host memory and
 * solve A x = b.  status BA_ERR_ZERO_PIVOT on an exactly zero pivot. */
int ba_dense_ldl_solve(int device, int64_t n, const double *a_lower_rowmajor, const double *b, double *x,
                       double *factor_ms);
/* the same with matrix and right-hand side rounded to Float32, factored and solved in Float32 (what
 * facto_type = Float32 does to the reduced camera system; src/lm.jl:170-173 does it to the augmented matrix) */
int ba_dense_ldl_solve_f32(int device, int64_t n, const double *a_lower_rowmajor, const double *b, double *x,
                           double *factor_ms);

#ifdef __cplusplus
}
#endif
#endif /* BA_HIP_H */
