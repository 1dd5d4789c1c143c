// TEST INFRASTRUCTURE (not part of the product library): an in-process, stream-ordered transport for the communication
// hook of include/ba_hip.h (ba_lm_set_comm_hook), so that ONE process on ONE GPU can run the multi-rank path of the LM
// step -- the reduce of the reduced camera matrix onto its owners, the distributed factorisation with its look-ahead, the
// panel broadcasts -- with genuinely asynchronous streams: every rank is a handle driven by its own host thread, and the
// three operations are device-to-device copies / a fixed-order sum kernel enqueued ON THE STREAM THE LIBRARY HANDS OVER,
// ordered across the ranks by events only.  No host thread ever waits for the GPU here (the gloo hook of
// bundleadjustment.jl_amd/parallel.py drains the stream on every call and so serialises the two streams of the
// look-ahead; real RCCL needs one GPU per rank).  The sums are formed in rank order, whatever the timing.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

namespace {

constexpr int MAXW = 8;

struct Loop {
  int world = 0;
  size_t cap = 0;
  void *stage[MAXW] = {};
  hipEvent_t ev_in[MAXW] = {}, ev_out[MAXW] = {};
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
  int arrived = 0;
  unsigned long generation = 0;
  bool broken = false;
  long ops = 0;
};
struct RankCtx {
  Loop *L;
  int rank;
};

// barrier of the `world` host threads; false when a rank failed to arrive within 120 s (its thread died): the hook then
// reports an error instead of hanging the test
bool rendezvous(Loop *L) {
  pthread_mutex_lock(&L->mu);
  if (L->broken) {
    pthread_mutex_unlock(&L->mu);
    return false;
  }
  const unsigned long gen = L->generation;
  if (++L->arrived == L->world) {
    L->arrived = 0;
    L->generation++;
    pthread_cond_broadcast(&L->cv);
    pthread_mutex_unlock(&L->mu);
    return true;
  }
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  ts.tv_sec += 120;
  while (gen == L->generation && !L->broken)
    if (pthread_cond_timedwait(&L->cv, &L->mu, &ts) != 0) {
      L->broken = true;
      pthread_cond_broadcast(&L->cv);
    }
  const bool ok = !L->broken;
  pthread_mutex_unlock(&L->mu);
  return ok;
}

struct Ptrs {
  const void *p[MAXW];
};

template <typename T>
__global__ void k_rank_order_sum(T *__restrict__ out, Ptrs src, int world, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    T s = static_cast<const T *>(src.p[0])[i];
    for (int r = 1; r < world; r++) s += static_cast<const T *>(src.p[r])[i];
    out[i] = s;
  }
}

#define LB_CHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "[loopback] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return 3;                                                                                 \
    }                                                                                           \
  } while (0)

}  // namespace

extern "C" void *ba_loopback_create(int world, size_t stage_bytes) {
  if (world < 1 || world > MAXW) return nullptr;
  Loop *L = new Loop();
  L->world = world;
  L->cap = stage_bytes;
  for (int r = 0; r < world; r++) {
    if (hipMalloc(&L->stage[r], stage_bytes ? stage_bytes : 1) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&L->ev_in[r], hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&L->ev_out[r], hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  (void)hipDeviceSynchronize();
  return L;
}

extern "C" void ba_loopback_destroy(void *h) {
  Loop *L = static_cast<Loop *>(h);
  if (!L) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < L->world; r++) {
    if (L->stage[r]) (void)hipFree(L->stage[r]);
    if (L->ev_in[r]) (void)hipEventDestroy(L->ev_in[r]);
    if (L->ev_out[r]) (void)hipEventDestroy(L->ev_out[r]);
  }
  delete L;
}

extern "C" void *ba_loopback_rank(void *h, int rank) {
  Loop *L = static_cast<Loop *>(h);
  if (!L || rank < 0 || rank >= L->world) return nullptr;
  return new RankCtx{L, rank};  // (a few bytes per rank and test: never freed)
}

extern "C" long ba_loopback_ops(void *h) { return h ? static_cast<Loop *>(h)->ops : -1; }

// the hook (ba_comm_fn).  op codes of include/ba_hip.h: 0 all-reduce f64, 1 reduce f64 onto root, 2 broadcast bytes, 3 reduce f32,
// 4 / 5 reduce-scatter f64 / f32 (in place: segment `rank` of the buffer receives the sum)
extern "C" int ba_loopback_hook(void *ctx, int op, void *d_buf, int64_t count, int root, void *stream) {
  RankCtx *c = static_cast<RankCtx *>(ctx);
  Loop *L = c->L;
  const int me = c->rank, W = L->world;
  hipStream_t st = (hipStream_t)stream;
  const bool rs = op == 4 || op == 5;  // reduce-scatter: `world` segments of `count` elements, segment `me` receives the sum
  const size_t esize = op == 2 ? 1 : ((op == 3 || op == 5) ? 4 : 8);
  const size_t bytes = (size_t)count * esize * (rs ? (size_t)W : 1);
  if (op < 0 || op > 5 || bytes > L->cap) {
    fprintf(stderr, "[loopback] op %d with %zu bytes: unsupported or beyond the staging buffers (%zu)\n", op, bytes, L->cap);
    return 2;
  }
  const bool bcast = op == 2, allred = op == 0;
  const bool sends = !bcast || me == root;
  const bool receives = allred || rs || (bcast ? me != root : me == root);
  // the previous operation's consumers are through with the staging buffers (their events were recorded before that
  // operation's second rendezvous, i.e. before any rank got here)
  for (int r = 0; r < W; r++) LB_CHECK(hipStreamWaitEvent(st, L->ev_out[r], 0));
  if (sends) LB_CHECK(hipMemcpyAsync(L->stage[me], d_buf, bytes, hipMemcpyDeviceToDevice, st));
  LB_CHECK(hipEventRecord(L->ev_in[me], st));
  if (!rendezvous(L)) return 4;  // every rank's contribution is enqueued and its event recorded
  if (receives) {
    if (bcast) {
      LB_CHECK(hipStreamWaitEvent(st, L->ev_in[root], 0));
      LB_CHECK(hipMemcpyAsync(d_buf, L->stage[root], bytes, hipMemcpyDeviceToDevice, st));
    } else {
      for (int r = 0; r < W; r++) LB_CHECK(hipStreamWaitEvent(st, L->ev_in[r], 0));
      Ptrs src;
      const size_t seg = rs ? (size_t)me * (size_t)count * esize : 0;  // reduce-scatter: this rank's segment of every buffer
      for (int r = 0; r < MAXW; r++) src.p[r] = r < W ? (const char *)L->stage[r] + seg : nullptr;
      const int64_t n = count;
      int64_t nb = (n + 255) / 256;
      if (nb > 4096) nb = 4096;
      if (esize == 4)
        hipLaunchKernelGGL(k_rank_order_sum<float>, dim3((unsigned)nb), dim3(256), 0, st, (float *)((char *)d_buf + seg), src, W, n);
      else
        hipLaunchKernelGGL(k_rank_order_sum<double>, dim3((unsigned)nb), dim3(256), 0, st, (double *)((char *)d_buf + seg), src, W, n);
      LB_CHECK(hipGetLastError());
    }
  }
  LB_CHECK(hipEventRecord(L->ev_out[me], st));
  if (me == 0) L->ops++;
  if (!rendezvous(L)) return 4;  // every consumer's event is recorded before anybody starts the next operation
  return 0;
}
