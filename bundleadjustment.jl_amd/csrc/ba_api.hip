// C ABI glue of libba_hip.so: errors, the problem handle (device mirrors of one BALNLPModel), the host-pointer
// NLPModels surface (cons! / jac_structure! / jac_coord! / J'r), device helpers, profiling.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "ba_internal.h"

static thread_local char g_err[512] = "";

void ba_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

const char *const kProfNames[PC_COUNT] = {
    "k_residual", "k_jac_structure", "k_jac_coord", "k_point_blocks", "k_cam_blocks", "k_schur_prep",
    "k_schur_blocks", "k_schur_rhs", "k_ldl_diag", "k_ldl_trsm", "k_ldl_col", "k_ldl_update", "k_ldl_update_rs", "k_tri_solve",
    "k_backsub", "k_model_sq", "k_reduce", "allreduce"};

extern "C" const char *ba_last_error(void) { return g_err; }

extern "C" int ba_device_count(int *n) {
  if (!n) return BA_ERR_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    ba_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return BA_ERR_HIP;
  }
  *n = c;
  return BA_OK;
}

extern "C" int ba_device_info(int dev, char *name, size_t name_cap, int *n_cu, size_t *hbm_bytes) {
  hipDeviceProp_t prop;
  BA_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return BA_OK;
}

int ba_scratch(ba_problem *p, int slot, size_t bytes, void **out) {
  if (p->scratch_bytes[slot] < bytes) {
    if (p->scratch[slot]) (void)hipFree(p->scratch[slot]);
    p->scratch[slot] = nullptr;
    p->scratch_bytes[slot] = 0;
    BA_HIP_CHECK(hipMalloc(&p->scratch[slot], bytes));
    p->scratch_bytes[slot] = bytes;
  }
  *out = p->scratch[slot];
  return BA_OK;
}

template <typename T>
static int upload(T **d, const std::vector<T> &h) {
  size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
  BA_HIP_CHECK(hipMalloc((void **)d, bytes));
  if (!h.empty()) BA_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return BA_OK;
}

extern "C" int ba_problem_create(int device, int64_t ncams, int64_t npnts, int64_t nobs, const int64_t *cam_idx1,
                                 const int64_t *pnt_idx1, const double *pt2d, ba_problem **out) {
  if (!out) return BA_ERR_ARG;
  *out = nullptr;
  if (ncams < 0 || npnts < 0 || nobs < 0 || (nobs > 0 && (!cam_idx1 || !pnt_idx1 || !pt2d))) {
    ba_set_error("ba_problem_create: bad sizes or null arrays");
    return BA_ERR_ARG;
  }
  if (nobs > (int64_t)88000000 || ncams > (int64_t)230000000 || npnts > (int64_t)700000000) {
    ba_set_error("ba_problem_create: problem exceeds the 32-bit device index range");
    return BA_ERR_ARG;
  }
  for (int64_t k = 0; k < nobs; k++) {
    if (cam_idx1[k] < 1 || cam_idx1[k] > ncams || pnt_idx1[k] < 1 || pnt_idx1[k] > npnts) {
      ba_set_error("ba_problem_create: observation %lld has index out of range (cam %lld of %lld, point %lld of %lld)",
                   (long long)(k + 1), (long long)cam_idx1[k], (long long)ncams, (long long)pnt_idx1[k],
                   (long long)npnts);
      return BA_ERR_ARG;
    }
  }
  int ndev = 0;
  BA_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    ba_set_error("ba_problem_create: device %d not available (%d devices)", device, ndev);
    return BA_ERR_HIP;
  }
  BA_HIP_CHECK(hipSetDevice(device));
  ba_problem *p = new ba_problem();
  p->device = device;
  p->ncams = ncams;
  p->npnts = npnts;
  p->nobs = nobs;
  p->h_cam0.resize((size_t)nobs);
  p->h_pnt0.resize((size_t)nobs);
  bool sorted = true;
  for (int64_t k = 0; k < nobs; k++) {
    p->h_cam0[(size_t)k] = (int)(cam_idx1[k] - 1);
    p->h_pnt0[(size_t)k] = (int)(pnt_idx1[k] - 1);
    if (k > 0 && pnt_idx1[k] < pnt_idx1[k - 1]) sorted = false;
  }
  p->point_sorted = sorted;
  // stable counting sorts: observations by point and by camera
  std::vector<int> cam_ptr((size_t)ncams + 1, 0), cam_obs((size_t)nobs);
  p->h_pt_ptr.assign((size_t)npnts + 1, 0);
  p->h_pt_obs.resize((size_t)nobs);
  for (int64_t k = 0; k < nobs; k++) {
    p->h_pt_ptr[(size_t)p->h_pnt0[(size_t)k] + 1]++;
    cam_ptr[(size_t)p->h_cam0[(size_t)k] + 1]++;
  }
  for (int64_t i = 0; i < npnts; i++) p->h_pt_ptr[(size_t)i + 1] += p->h_pt_ptr[(size_t)i];
  for (int64_t i = 0; i < ncams; i++) cam_ptr[(size_t)i + 1] += cam_ptr[(size_t)i];
  {
    std::vector<int> cp(p->h_pt_ptr.begin(), p->h_pt_ptr.end() - 1), cc(cam_ptr.begin(), cam_ptr.end() - 1);
    for (int64_t k = 0; k < nobs; k++) {
      p->h_pt_obs[(size_t)cp[(size_t)p->h_pnt0[(size_t)k]]++] = (int)k;
      cam_obs[(size_t)cc[(size_t)p->h_cam0[(size_t)k]]++] = (int)k;
    }
  }
  int rc = BA_OK;
  do {
    if ((rc = upload(&p->cam0, p->h_cam0)) != BA_OK) break;
    if ((rc = upload(&p->pnt0, p->h_pnt0)) != BA_OK) break;
    if ((rc = upload(&p->pt_ptr, p->h_pt_ptr)) != BA_OK) break;
    if ((rc = upload(&p->pt_obs, p->h_pt_obs)) != BA_OK) break;
    if ((rc = upload(&p->cam_ptr, cam_ptr)) != BA_OK) break;
    if ((rc = upload(&p->cam_obs, cam_obs)) != BA_OK) break;
    std::vector<double> h2((size_t)2 * nobs);
    std::vector<float> h2f((size_t)2 * nobs);
    for (int64_t k = 0; k < 2 * nobs; k++) {
      h2[(size_t)k] = pt2d[k];
      h2f[(size_t)k] = (float)pt2d[k];
    }
    if ((rc = upload(&p->pt2d, h2)) != BA_OK) break;
    if ((rc = upload(&p->pt2d_f32, h2f)) != BA_OK) break;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e != hipSuccess) {
      ba_set_error("ba_problem_create: %s", hipGetErrorString(e));
      rc = BA_ERR_HIP;
    }
  } while (0);
  if (rc != BA_OK) {
    ba_problem_destroy(p);
    return rc;
  }
  *out = p;
  return BA_OK;
}

extern "C" void ba_problem_destroy(ba_problem *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  lm_free(p);
  comm_free(p);
  void *ptrs[] = {p->cam0, p->pnt0, p->pt2d, p->pt2d_f32, p->pt_ptr, p->pt_obs, p->cam_ptr, p->cam_obs,
                  p->scratch[0], p->scratch[1], p->scratch[2], p->scratch[3]};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

extern "C" int ba_problem_dims(const ba_problem *p, int64_t *ncams, int64_t *npnts, int64_t *nobs, int64_t *nvar,
                               int64_t *nequ, int64_t *nnzj) {
  if (!p) return BA_ERR_ARG;
  if (ncams) *ncams = p->ncams;
  if (npnts) *npnts = p->npnts;
  if (nobs) *nobs = p->nobs;
  if (nvar) *nvar = 9 * p->ncams + 3 * p->npnts;  // src/BALNLPModels.jl:95
  if (nequ) *nequ = 2 * p->nobs;                  // :97
  if (nnzj) *nnzj = 24 * p->nobs;                 // :102
  return BA_OK;
}

static inline hipStream_t pick(ba_problem *p, void *stream) { return stream ? (hipStream_t)stream : p->stream; }

#define NEED(p, ...)                                   \
  do {                                                 \
    if (!(p)) {                                        \
      ba_set_error("%s: null handle", __func__);       \
      return BA_ERR_ARG;                               \
    }                                                  \
    const void *_a[] = {__VA_ARGS__};                  \
    for (const void *q : _a)                           \
      if (!q && (p)->nobs > 0) {                       \
        ba_set_error("%s: null array", __func__);      \
        return BA_ERR_ARG;                             \
      }                                                \
    BA_HIP_CHECK(hipSetDevice((p)->device));           \
  } while (0)

// ---- device-resident entries ------------------------------------------------------------------------------------------
extern "C" int ba_residual_dev(ba_problem *p, const double *d_x, double *d_r, void *stream) {
  NEED(p, d_x, d_r);
  return launch_residual_f64(p, d_x, d_r, pick(p, stream));
}
extern "C" int ba_residual_f32_dev(ba_problem *p, const float *d_x, float *d_r, void *stream) {
  NEED(p, d_x, d_r);
  return launch_residual_f32(p, d_x, d_r, pick(p, stream));
}
extern "C" int ba_jac_structure_dev(ba_problem *p, int64_t *d_rows, int64_t *d_cols, void *stream) {
  NEED(p, d_rows, d_cols);
  return launch_jac_structure(p, d_rows, d_cols, pick(p, stream));
}
extern "C" int ba_jac_coord_dev(ba_problem *p, const double *d_x, double *d_vals, void *stream) {
  NEED(p, d_x, d_vals);
  return launch_jac_coord_f64(p, d_x, d_vals, pick(p, stream));
}
extern "C" int ba_jac_coord_f32_dev(ba_problem *p, const float *d_x, float *d_vals, void *stream) {
  NEED(p, d_x, d_vals);
  return launch_jac_coord_f32(p, d_x, d_vals, pick(p, stream));
}
extern "C" int ba_jtr_dev(ba_problem *p, const double *d_vals, const double *d_r, double *d_jtr, void *stream) {
  NEED(p, d_vals, d_r, d_jtr);
  hipStream_t st = pick(p, stream);
  BA_CHECK(launch_point_blocks(p, d_vals, d_r, nullptr, d_jtr, st));
  BA_CHECK(launch_cam_blocks(p, d_vals, d_r, nullptr, d_jtr + 3 * p->npnts, st));
  return BA_OK;
}

// ---- host-pointer entries: copy in, run, copy out, synchronise -----------------------------------------------------------
template <typename T>
static int host_residual(ba_problem *p, const T *x, T *r) {
  const int64_t nvar = 9 * p->ncams + 3 * p->npnts;
  T *dx, *dr;
  BA_CHECK(ba_scratch(p, 0, (size_t)(nvar + 1) * sizeof(T), (void **)&dx));
  BA_CHECK(ba_scratch(p, 1, (size_t)(2 * p->nobs + 1) * sizeof(T), (void **)&dr));
  BA_HIP_CHECK(hipMemcpyAsync(dx, x, (size_t)nvar * sizeof(T), hipMemcpyHostToDevice, p->stream));
  if constexpr (sizeof(T) == 8) BA_CHECK(launch_residual_f64(p, (const double *)dx, (double *)dr, p->stream));
  else BA_CHECK(launch_residual_f32(p, (const float *)dx, (float *)dr, p->stream));
  BA_HIP_CHECK(hipMemcpyAsync(r, dr, (size_t)2 * p->nobs * sizeof(T), hipMemcpyDeviceToHost, p->stream));
  BA_HIP_CHECK(hipStreamSynchronize(p->stream));
  return BA_OK;
}

extern "C" int ba_residual(ba_problem *p, const double *x, double *r) {
  NEED(p, x, r);
  return host_residual<double>(p, x, r);
}
extern "C" int ba_residual_f32(ba_problem *p, const float *x, float *r) {
  NEED(p, x, r);
  return host_residual<float>(p, x, r);
}

extern "C" int ba_jac_structure(ba_problem *p, int64_t *rows, int64_t *cols) {
  NEED(p, rows, cols);
  const size_t bytes = (size_t)24 * p->nobs * sizeof(int64_t);
  int64_t *dr, *dc;
  BA_CHECK(ba_scratch(p, 0, bytes + 16, (void **)&dr));
  BA_CHECK(ba_scratch(p, 1, bytes + 16, (void **)&dc));
  BA_CHECK(launch_jac_structure(p, dr, dc, p->stream));
  BA_HIP_CHECK(hipMemcpyAsync(rows, dr, bytes, hipMemcpyDeviceToHost, p->stream));
  BA_HIP_CHECK(hipMemcpyAsync(cols, dc, bytes, hipMemcpyDeviceToHost, p->stream));
  BA_HIP_CHECK(hipStreamSynchronize(p->stream));
  return BA_OK;
}

template <typename T>
static int host_jac_coord(ba_problem *p, const T *x, T *vals) {
  const int64_t nvar = 9 * p->ncams + 3 * p->npnts;
  T *dx, *dv;
  BA_CHECK(ba_scratch(p, 0, (size_t)(nvar + 1) * sizeof(T), (void **)&dx));
  BA_CHECK(ba_scratch(p, 1, (size_t)(24 * p->nobs + 2) * sizeof(T), (void **)&dv));
  BA_HIP_CHECK(hipMemcpyAsync(dx, x, (size_t)nvar * sizeof(T), hipMemcpyHostToDevice, p->stream));
  if constexpr (sizeof(T) == 8) BA_CHECK(launch_jac_coord_f64(p, (const double *)dx, (double *)dv, p->stream));
  else BA_CHECK(launch_jac_coord_f32(p, (const float *)dx, (float *)dv, p->stream));
  BA_HIP_CHECK(hipMemcpyAsync(vals, dv, (size_t)24 * p->nobs * sizeof(T), hipMemcpyDeviceToHost, p->stream));
  BA_HIP_CHECK(hipStreamSynchronize(p->stream));
  return BA_OK;
}

extern "C" int ba_jac_coord(ba_problem *p, const double *x, double *vals) {
  NEED(p, x, vals);
  return host_jac_coord<double>(p, x, vals);
}
extern "C" int ba_jac_coord_f32(ba_problem *p, const float *x, float *vals) {
  NEED(p, x, vals);
  return host_jac_coord<float>(p, x, vals);
}

extern "C" int ba_jtr(ba_problem *p, const double *vals, const double *r, double *jtr) {
  NEED(p, vals, r, jtr);
  const int64_t nvar = 9 * p->ncams + 3 * p->npnts;
  double *dv, *dr, *dj;
  BA_CHECK(ba_scratch(p, 0, (size_t)(24 * p->nobs + 2) * sizeof(double), (void **)&dv));
  BA_CHECK(ba_scratch(p, 1, (size_t)(2 * p->nobs + 2) * sizeof(double), (void **)&dr));
  BA_CHECK(ba_scratch(p, 2, (size_t)(nvar + 1) * sizeof(double), (void **)&dj));
  BA_HIP_CHECK(hipMemcpyAsync(dv, vals, (size_t)24 * p->nobs * sizeof(double), hipMemcpyHostToDevice, p->stream));
  BA_HIP_CHECK(hipMemcpyAsync(dr, r, (size_t)2 * p->nobs * sizeof(double), hipMemcpyHostToDevice, p->stream));
  BA_CHECK(ba_jtr_dev(p, dv, dr, dj, p->stream));
  BA_HIP_CHECK(hipMemcpyAsync(jtr, dj, (size_t)nvar * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  BA_HIP_CHECK(hipStreamSynchronize(p->stream));
  return BA_OK;
}

// ---- device helpers --------------------------------------------------------------------------------------------------------
extern "C" int ba_dev_malloc(ba_problem *p, size_t bytes, void **d_ptr) {
  if (!p || !d_ptr) return BA_ERR_ARG;
  BA_HIP_CHECK(hipSetDevice(p->device));
  BA_HIP_CHECK(hipMalloc(d_ptr, bytes ? bytes : 1));
  return BA_OK;
}
extern "C" int ba_dev_free(ba_problem *p, void *d_ptr) {
  if (!p) return BA_ERR_ARG;
  BA_HIP_CHECK(hipSetDevice(p->device));
  if (d_ptr) BA_HIP_CHECK(hipFree(d_ptr));
  return BA_OK;
}
// Copies between host and device memory are ORDERED ON A STREAM and complete on return: the copy is enqueued on `stream`
// (null: the handle's own stream), behind everything enqueued there before, and the call returns when that stream has
// drained.  A null-stream hipMemcpy has no ordering against the library's hipStreamNonBlocking streams: data it writes is
// not ordered before (nor made visible to) kernels launched afterwards on a stream the host never synchronised.
static int copy_on(ba_problem *p, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, void *stream) {
  if (!p || (bytes && (!dst || !src))) return BA_ERR_ARG;
  BA_HIP_CHECK(hipSetDevice(p->device));
  hipStream_t st = stream ? (hipStream_t)stream : p->stream;
  if (bytes) BA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, st));
  BA_HIP_CHECK(hipStreamSynchronize(st));
  return BA_OK;
}
extern "C" int ba_memcpy_h2d(ba_problem *p, void *d_dst, const void *h_src, size_t bytes) {
  return copy_on(p, d_dst, h_src, bytes, hipMemcpyHostToDevice, nullptr);
}
extern "C" int ba_memcpy_d2h(ba_problem *p, void *h_dst, const void *d_src, size_t bytes) {
  return copy_on(p, h_dst, d_src, bytes, hipMemcpyDeviceToHost, nullptr);
}
extern "C" int ba_memcpy_h2d_on(ba_problem *p, void *stream, void *d_dst, const void *h_src, size_t bytes) {
  return copy_on(p, d_dst, h_src, bytes, hipMemcpyHostToDevice, stream);
}
extern "C" int ba_memcpy_d2h_on(ba_problem *p, void *stream, void *h_dst, const void *d_src, size_t bytes) {
  return copy_on(p, h_dst, d_src, bytes, hipMemcpyDeviceToHost, stream);
}
extern "C" int ba_synchronize(ba_problem *p) {
  if (!p) return BA_ERR_ARG;
  BA_HIP_CHECK(hipSetDevice(p->device));
  BA_HIP_CHECK(hipStreamSynchronize(p->stream));
  return BA_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------------------------------
extern "C" int ba_profile_enable(ba_problem *p, int on) {
  if (!p) return BA_ERR_ARG;
  p->prof_on = on != 0;
  return BA_OK;
}
extern "C" int ba_profile_reset(ba_problem *p) {
  if (!p) return BA_ERR_ARG;
  for (auto &s : p->prof) s = ProfSlot();
  return BA_OK;
}
extern "C" int ba_profile_get(ba_problem *p, int cap, const char **names, double *total_ms, int64_t *calls, int *n) {
  if (!p || !n) return BA_ERR_ARG;
  *n = PC_COUNT;
  for (int i = 0; i < PC_COUNT && i < cap; i++) {
    if (names) names[i] = kProfNames[i];
    if (total_ms) total_ms[i] = p->prof[i].ms;
    if (calls) calls[i] = p->prof[i].calls;
  }
  return BA_OK;
}

// src/lm.jl:84-88 (`perm`): the camera ordering of a problem and the tile fill it leaves -- host only (ba_order.cpp)
int schur_ordering_host(int64_t ncams, int64_t npnts, int64_t nobs, const int64_t *cam_idx1, const int64_t *pnt_idx1, int method,
                        int nb, int64_t *perm1, double *tile_fill, double *flop_fill, double *block_fill);
extern "C" int ba_schur_ordering(int64_t ncams, int64_t npnts, int64_t nobs, const int64_t *cam_idx1, const int64_t *pnt_idx1,
                                 int method, int64_t *perm1, double *tile_fill, double *flop_fill, double *block_fill) {
  if (method < 0 || method > 2) {
    ba_set_error("ba_schur_ordering: method must be 0 (:AMD), 1 (:Metis) or 2 (the caller's numbering)");
    return BA_ERR_ARG;
  }
  const int rc = schur_ordering_host(ncams, npnts, nobs, cam_idx1, pnt_idx1, method, NB, perm1, tile_fill, flop_fill, block_fill);
  if (rc != 0) {
    ba_set_error(rc == 1 ? "ba_schur_ordering: bad sizes or index out of range" : "ba_schur_ordering: internal error (incomplete sequence)");
    return BA_ERR_ARG;
  }
  return BA_OK;
}
