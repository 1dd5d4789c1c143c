// Cross-rank transport of the multi-GPU LM (one process per GPU, observations sharded by point).
//
// The reference is single-process: nothing of this has a counterpart there.  What travels between ranks per LM
// iteration is only camera-side data:
//   * all-reduce of short Float64 vectors (J'r camera part, diag(J'J), the right-hand side, a few scalars),
//   * reduce of each owner's contiguous range of the reduced camera matrix S onto that owner,
//   * broadcast of the factored panels (V = L D, the inverted diagonal tiles, the pivots) from their owner.
// Two transports implement the three operations:
//   * RCCL, called directly from here on the library's stream (librccl.so.1 is dlopen'ed: a single-GPU user needs no
//     RCCL; in a process that already holds RCCL -- e.g. torch -- the same instance is reused): ba_lm_set_comm_rccl;
//   * a caller-supplied hook (ba_lm_set_comm_hook): lets a host language carry the data over whatever it has (MPI.jl,
//     gloo in the CPU-side tests, several ranks sharing one GPU where RCCL cannot run).
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#include "ba_internal.h"

namespace {

struct RcclApi {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};

RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.dl) return BA_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *dl = nullptr;
  for (const char *n : names) {
    dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (dl) break;
  }
  if (!dl) {
    ba_set_error("RCCL runtime not found (librccl.so.1): %s", dlerror());
    return BA_ERR_COMM;
  }
#define BA_SYM(field, name)                                               \
  *(void **)(&g_rccl.field) = dlsym(dl, name);                             \
  if (!g_rccl.field) {                                                     \
    ba_set_error("RCCL runtime lacks %s", name);                           \
    dlclose(dl);                                                           \
    return BA_ERR_COMM;                                                    \
  }
  BA_SYM(GetUniqueId, "ncclGetUniqueId")
  BA_SYM(CommInitRank, "ncclCommInitRank")
  BA_SYM(CommDestroy, "ncclCommDestroy")
  BA_SYM(GetErrorString, "ncclGetErrorString")
  BA_SYM(AllReduce, "ncclAllReduce")
  BA_SYM(Reduce, "ncclReduce")
  BA_SYM(Broadcast, "ncclBroadcast")
  BA_SYM(ReduceScatter, "ncclReduceScatter")
  BA_SYM(GroupStart, "ncclGroupStart")
  BA_SYM(GroupEnd, "ncclGroupEnd")
#undef BA_SYM
  g_rccl.dl = dl;
  return BA_OK;
}

#define BA_NCCL_CHECK(expr)                                                                   \
  do {                                                                                        \
    ncclResult_t _r = (expr);                                                                 \
    if (_r != ncclSuccess) {                                                                  \
      ba_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));  \
      return BA_ERR_COMM;                                                                     \
    }                                                                                         \
  } while (0)

int hook_call(BaComm *c, int op, void *buf, int64_t count, int root, hipStream_t st) {
  int rc = c->hook(c->hook_ctx, op, buf, count, root, (void *)st);
  if (rc != 0) {
    ba_set_error("communication hook failed (op %d, rc %d)", op, rc);
    return BA_ERR_COMM;
  }
  return BA_OK;
}

}  // namespace

int comm_allreduce(ba_problem *p, double *d_buf, int64_t count, hipStream_t st) {
  BaComm *c = &p->comm;
  if (!c->active() || count <= 0) return BA_OK;
  ProfScope ps(p, PC_COMM, st);
  c->calls++;
  c->bytes += 8 * count;
  c->op_calls[BA_COMM_ALLREDUCE_F64]++;
  c->op_bytes[BA_COMM_ALLREDUCE_F64] += 8 * count;
  if (c->hook) return hook_call(c, BA_COMM_ALLREDUCE_F64, d_buf, count, 0, st);
  BA_NCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)c->nccl, st));
  return BA_OK;
}

int comm_reduce(ba_problem *p, double *d_buf, int64_t count, int root, hipStream_t st) {
  BaComm *c = &p->comm;
  if (!c->active() || count <= 0) return BA_OK;
  ProfScope ps(p, PC_COMM, st);
  c->calls++;
  c->bytes += 8 * count;
  c->op_calls[BA_COMM_REDUCE_F64]++;
  c->op_bytes[BA_COMM_REDUCE_F64] += 8 * count;
  if (c->hook) return hook_call(c, BA_COMM_REDUCE_F64, d_buf, count, root, st);
  BA_NCCL_CHECK(g_rccl.Reduce(d_buf, d_buf, (size_t)count, ncclFloat64, ncclSum, root, (ncclComm_t)c->nccl, st));
  return BA_OK;
}

int comm_reduce_f32(ba_problem *p, float *d_buf, int64_t count, int root, hipStream_t st) {
  BaComm *c = &p->comm;
  if (!c->active() || count <= 0) return BA_OK;
  ProfScope ps(p, PC_COMM, st);
  c->calls++;
  c->bytes += 4 * count;
  c->op_calls[BA_COMM_REDUCE_F32]++;
  c->op_bytes[BA_COMM_REDUCE_F32] += 4 * count;
  if (c->hook) return hook_call(c, BA_COMM_REDUCE_F32, d_buf, count, root, st);
  BA_NCCL_CHECK(g_rccl.Reduce(d_buf, d_buf, (size_t)count, ncclFloat32, ncclSum, root, (ncclComm_t)c->nccl, st));
  return BA_OK;
}

int comm_bcast(ba_problem *p, void *d_buf, int64_t bytes, int root, hipStream_t st) {
  BaComm *c = &p->comm;
  if (!c->active() || bytes <= 0) return BA_OK;
  ProfScope ps(p, PC_COMM, st);
  c->calls++;
  c->bytes += bytes;
  c->op_calls[BA_COMM_BCAST_BYTES]++;
  c->op_bytes[BA_COMM_BCAST_BYTES] += bytes;
  if (c->hook) return hook_call(c, BA_COMM_BCAST_BYTES, d_buf, bytes, root, st);
  BA_NCCL_CHECK(g_rccl.Broadcast(d_buf, d_buf, (size_t)bytes, ncclUint8, root, (ncclComm_t)c->nccl, st));
  return BA_OK;
}

int comm_reduce_scatter(ba_problem *p, void *d_buf, int64_t count, bool f32, hipStream_t st) {
  BaComm *c = &p->comm;
  if (!c->active() || count <= 0) return BA_OK;
  ProfScope ps(p, PC_COMM, st);
  const int op = f32 ? BA_COMM_REDUCE_SCATTER_F32 : BA_COMM_REDUCE_SCATTER_F64;
  const int64_t esize = f32 ? 4 : 8, bytes = esize * count * c->world;
  c->calls++;
  c->bytes += bytes;
  c->op_calls[op]++;
  c->op_bytes[op] += bytes;
  if (c->hook) return hook_call(c, op, d_buf, count, 0, st);
  // in place: the receive buffer is this rank's own segment of the send buffer
  BA_NCCL_CHECK(g_rccl.ReduceScatter(d_buf, (char *)d_buf + (int64_t)c->rank * count * esize, (size_t)count, f32 ? ncclFloat32 : ncclFloat64,
                                     ncclSum, (ncclComm_t)c->nccl, st));
  return BA_OK;
}

int comm_group_begin(ba_problem *p) {
  BaComm *c = &p->comm;
  if (c->active() && !c->hook) BA_NCCL_CHECK(g_rccl.GroupStart());
  return BA_OK;
}

int comm_group_end(ba_problem *p) {
  BaComm *c = &p->comm;
  if (c->active() && !c->hook) BA_NCCL_CHECK(g_rccl.GroupEnd());
  return BA_OK;
}

void comm_free(ba_problem *p) {
  BaComm *c = &p->comm;
  if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->nccl);
  *c = BaComm();
}

static int comm_args_ok(ba_problem *p, int rank, int world) {
  if (!p || world < 1 || rank < 0 || rank >= world) {
    ba_set_error("communicator: bad rank / world (%d / %d)", rank, world);
    return BA_ERR_ARG;
  }
  if (p->lm) {
    ba_set_error("the communicator must be set before the first solve on this handle");
    return BA_ERR_ARG;
  }
  return BA_OK;
}

extern "C" int ba_lm_set_comm_hook(ba_problem *p, int rank, int world, ba_comm_fn fn, void *ctx) {
  BA_CHECK(comm_args_ok(p, rank, world));
  if (world > 1 && !fn) {
    ba_set_error("ba_lm_set_comm_hook: world > 1 needs a hook");
    return BA_ERR_ARG;
  }
  comm_free(p);
  p->comm.rank = rank;
  p->comm.world = world;
  p->comm.hook = fn;
  p->comm.hook_ctx = ctx;
  p->rank = rank;
  p->world = world;
  return BA_OK;
}

extern "C" int ba_comm_get_unique_id(void *id_out) {
  if (!id_out) return BA_ERR_ARG;
  BA_CHECK(rccl_load());
  ncclUniqueId id;
  BA_NCCL_CHECK(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(ncclUniqueId) == BA_COMM_ID_BYTES, "BA_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
  memcpy(id_out, &id, sizeof id);
  return BA_OK;
}

extern "C" int ba_lm_set_comm_rccl(ba_problem *p, int rank, int world, const void *id_in) {
  BA_CHECK(comm_args_ok(p, rank, world));
  if (!id_in) {
    ba_set_error("ba_lm_set_comm_rccl: null unique id");
    return BA_ERR_ARG;
  }
  BA_CHECK(rccl_load());
  BA_HIP_CHECK(hipSetDevice(p->device));
  comm_free(p);
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof id);
  ncclComm_t comm = nullptr;
  BA_NCCL_CHECK(g_rccl.CommInitRank(&comm, world, id, rank));
  p->comm.rank = rank;
  p->comm.world = world;
  p->comm.nccl = (void *)comm;
  p->rank = rank;
  p->world = world;
  return BA_OK;
}

extern "C" int ba_comm_stats(ba_problem *p, int64_t *calls, int64_t *bytes) {
  if (!p) return BA_ERR_ARG;
  if (calls) *calls = p->comm.calls;
  if (bytes) *bytes = p->comm.bytes;
  return BA_OK;
}

extern "C" int ba_comm_stats_ops(ba_problem *p, int64_t *calls, int64_t *bytes) {
  if (!p) return BA_ERR_ARG;
  for (int q = 0; q < BA_COMM_OPS; q++) {
    if (calls) calls[q] = p->comm.op_calls[q];
    if (bytes) bytes[q] = p->comm.op_bytes[q];
  }
  return BA_OK;
}

// the distribution of the reduced camera matrix, for hosts and tests (pure host code, needs no device)
extern "C" int ba_dist_layout(int64_t nt, int world, int64_t *col_off, int64_t *own_range) {
  if (nt < 1 || world < 1 || !col_off) return BA_ERR_ARG;
  std::vector<int64_t> co, own;
  dense_ldl_layout(nt, world, &co, &own);
  for (int64_t j = 0; j < nt; j++) col_off[j] = co[(size_t)j];
  if (own_range)
    for (int r = 0; r <= world; r++) own_range[r] = own[(size_t)r];
  return BA_OK;
}
