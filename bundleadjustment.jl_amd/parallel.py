"""Multi-GPU sharding of one bundle-adjustment problem: observations are partitioned by point (each observation
touches one point block and one camera block, src/BALNLPModels.jl:137-139; BAL files list observations grouped by
point), cameras are replicated.  Residual, Jacobian and all point-side blocks are then rank-local; only camera-side
sums (J'r camera part, the reduced camera matrix S and its right-hand side) and a few scalars are summed across
ranks, by torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).
The reference itself is single-process (no collective exists in it).
"""
import ctypes as C

import numpy as np


def partition_by_point(pnt_idx1, npnts, world):
    """Contiguous point ranges balanced by observation count.
    -> list of (p_begin, p_end) 0-based half-open point ranges, one per rank."""
    pnt0 = np.asarray(pnt_idx1, dtype=np.int64) - 1
    deg = np.bincount(pnt0, minlength=npnts)
    csum = np.concatenate([[0], np.cumsum(deg)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), npnts)
        bounds.append(b)
    bounds.append(npnts)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def shard_problem(arrays, rank, world):
    """arrays = (cam_idx1, pnt_idx1, pt2d, x0, ncams, npnts, nobs) as readfile() returns them.
    -> (local arrays in the same order with points renumbered 1..npnts_local, info dict)."""
    cam, pnt, pt2d, x0, ncams, npnts, nobs = arrays
    cam = np.asarray(cam, dtype=np.int64)
    pnt = np.asarray(pnt, dtype=np.int64)
    pb, pe = partition_by_point(pnt, npnts, world)[rank]
    sel = np.flatnonzero((pnt - 1 >= pb) & (pnt - 1 < pe))
    cam_l = np.ascontiguousarray(cam[sel])
    pnt_l = np.ascontiguousarray(pnt[sel] - pb)
    pt2d_l = np.ascontiguousarray(np.asarray(pt2d).reshape(-1, 2)[sel].ravel())
    x0 = np.asarray(x0)
    x0_l = np.concatenate([x0[3 * pb: 3 * pe], x0[3 * npnts:]])
    info = dict(point_range=(pb, pe), obs_index=sel, npnts_global=npnts, nobs_global=nobs)
    return (cam_l, pnt_l, pt2d_l, x0_l, ncams, pe - pb, len(sel)), info


def gather_solution(x_local, info, ncams, group=None):
    """Reassemble the global x = [points; cameras] from the shards (cameras are identical on every rank)."""
    import torch
    import torch.distributed as dist
    pb, pe = info["point_range"]
    npnts = info["npnts_global"]
    pts = torch.zeros(3 * npnts, dtype=torch.float64)
    pts[3 * pb: 3 * pe] = torch.from_numpy(np.ascontiguousarray(x_local[: 3 * (pe - pb)]))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            pts = pts.cuda()
            dist.all_reduce(pts, group=group)
            pts = pts.cpu()
        else:
            dist.all_reduce(pts, group=group)
    return np.concatenate([pts.numpy(), x_local[3 * (pe - pb):]])


class CameraBlockReducer:
    """Attaches the cross-rank transport of one shard's handle (include/ba_hip.h, "multi-GPU"): the camera-side sums
    (J'r camera part, diag(J'J), right-hand side, scalars: all-reduce), the reduce of each rank's part of the reduced
    camera matrix onto its owner, and the broadcast of the factored panels.

    * torch.distributed backend "nccl" (= RCCL over xGMI on MI355X): rank 0 draws the RCCL unique id from the library,
      torch.distributed only carries those 128 bytes to the other ranks; the library then opens its OWN RCCL communicator
      (ba_lm_set_comm_rccl) and issues every collective itself, from C, on its stream.  No Python in the LM loop.
    * backend "gloo" (the CPU-side tests; several ranks sharing one GPU, where RCCL cannot run): the hook transport
      (ba_lm_set_comm_hook) -- each operation is staged through host memory and carried by gloo."""

    def __init__(self, nlp, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib, self.nlp = _lib, nlp
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        L = _lib.lib()
        if self.backend == "nccl":
            ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
                _lib.check(L.ba_comm_get_unique_id(buf))
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            ident = ident.to(f"cuda:{nlp.device}")
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            raw = bytes(ident.cpu().tolist())
            self._id = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(raw)
            _lib.check(L.ba_lm_set_comm_rccl(nlp.handle, self.rank, self.world, self._id))
            self._cb = None
            return

        h = nlp.handle

        def _hook(ctx, op, d_buf, count, root, stream):
            # Every copy is enqueued ON THE STREAM THE LIBRARY HANDS OVER (ba_memcpy_*_on) and that stream is drained: the
            # operation is ordered behind the producer of d_buf and ahead of what the library launches next on `stream`
            # (include/ba_hip.h).  `stream` is the transfer stream of the distributed factorisation's look-ahead as often as
            # the handle's main stream; both are non-blocking, a null-stream copy has no ordering against either.
            try:
                esize, view = {_lib.COMM_BCAST_BYTES: (1, np.uint8), _lib.COMM_REDUCE_F32: (4, np.float32),
                               _lib.COMM_REDUCE_SCATTER_F32: (4, np.float32)}.get(op, (8, np.float64))
                scatter = op in (_lib.COMM_REDUCE_SCATTER_F64, _lib.COMM_REDUCE_SCATTER_F32)
                nbytes = esize * count * (self.world if scatter else 1)
                host = np.empty(nbytes, dtype=np.uint8)
                st = C.c_void_p(stream)
                sends = not (op == _lib.COMM_BCAST_BYTES and self.rank != root)  # a broadcast's receivers have nothing to read
                if sends:
                    _lib.check(L.ba_memcpy_d2h_on(h, st, _lib.ptr(host), C.c_void_p(d_buf), nbytes))
                t = torch.from_numpy(host.view(view))
                src = (lambda r: dist.get_global_rank(group, r) if group is not None else r)
                if op == _lib.COMM_ALLREDUCE_F64:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                elif op in (_lib.COMM_REDUCE_F64, _lib.COMM_REDUCE_F32):
                    dist.reduce(t, dst=src(root), op=dist.ReduceOp.SUM, group=group)
                    if self.rank != root:
                        return 0  # only the root's buffer receives the sum
                elif op == _lib.COMM_BCAST_BYTES:
                    dist.broadcast(t, src=src(root), group=group)
                    if self.rank == root:
                        return 0
                elif scatter:
                    # gloo has no reduce-scatter: the whole buffer is summed and this rank's segment goes back (a test
                    # transport: the bytes on the wire are not what RCCL's reduce-scatter moves)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    seg = esize * count
                    _lib.check(L.ba_memcpy_h2d_on(h, st, C.c_void_p(d_buf + self.rank * seg), _lib.ptr(host[self.rank * seg:(self.rank + 1) * seg]), seg))
                    return 0
                else:
                    return 2
                _lib.check(L.ba_memcpy_h2d_on(h, st, C.c_void_p(d_buf), _lib.ptr(host), nbytes))
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import sys
                print(f"[ba] communication hook failed: {e!r}", file=sys.stderr)
                return 1

        self._cb = _lib.COMM_CB(_hook)  # keep alive as long as the handle may call it
        nlp._comm_hook_owner = self    # ... i.e. as long as the model lives, whether or not the caller keeps this object
        _lib.check(L.ba_lm_set_comm_hook(nlp.handle, self.rank, self.world, self._cb, None))

    def _stats(self):
        calls, nbytes = C.c_int64(0), C.c_int64(0)
        self._lib.check(self._lib.lib().ba_comm_stats(self.nlp.handle, C.byref(calls), C.byref(nbytes)))
        return calls.value, nbytes.value

    def stats_by_op(self):
        """{operation: (calls, bytes handed to the transport by this rank)} since the communicator was attached"""
        calls = (C.c_int64 * self._lib.COMM_OPS)()
        nbytes = (C.c_int64 * self._lib.COMM_OPS)()
        self._lib.check(self._lib.lib().ba_comm_stats_ops(self.nlp.handle, calls, nbytes))
        return {self._lib.COMM_OP_NAMES[q]: (int(calls[q]), int(nbytes[q])) for q in range(self._lib.COMM_OPS) if calls[q]}

    @property
    def calls(self):
        return self._stats()[0]

    @property
    def bytes(self):
        return self._stats()[1]


def dist_layout(nt, world):
    """(col_off, own_range) of the reduced camera matrix over `world` ranks (ba_dist_layout)."""
    from . import _lib
    col_off = np.zeros(nt, dtype=np.int64)
    own = np.zeros(world + 1, dtype=np.int64)
    _lib.check(_lib.lib().ba_dist_layout(nt, world, _lib.ptr(col_off), _lib.ptr(own)))
    return col_off, own


def allreduce_sum_numpy(a, group=None):
    """Sum a numpy float64 array over the ranks (gloo on CPU); used by the CPU tests of the sharding logic."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()
