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
    """Owns the reduce buffer of one shard (a torch tensor, so that torch.distributed can address it) and the
    all-reduce hook the C library calls (ba_lm_set_comm).  The hook sums buf[offset : offset+count] over the ranks in
    place.  With the NCCL (= RCCL) backend the collective is enqueued on the stream the library passes in; with gloo
    (CPU tests, or several ranks sharing one GPU) the range is staged through pinned host memory."""

    def __init__(self, nlp, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        n = C.c_int64(0)
        _lib.check(_lib.lib().ba_lm_reduce_doubles(nlp.handle, C.byref(n)))
        self.buf = torch.zeros(n.value, dtype=torch.float64, device=f"cuda:{nlp.device}")
        self.calls = 0
        self.bytes = 0
        self._host = None

        def _hook(ctx, offset, count, stream):
            try:
                view = self.buf[offset: offset + count]
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
                if self.backend == "nccl":
                    with torch.cuda.stream(ext):
                        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    # host staging, fully synchronous, on torch's own stream (torch's pinned-memory allocator must not
                    # record events on the library's stream: that stream dies with the handle)
                    ext.synchronize()
                    if self._host is None or self._host.numel() < count:
                        self._host = torch.empty(max(count, 1 << 16), dtype=torch.float64).pin_memory()
                    h = self._host[:count]
                    h.copy_(view)
                    torch.cuda.current_stream().synchronize()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                    view.copy_(h)
                    torch.cuda.current_stream().synchronize()
                self.calls += 1
                self.bytes += 8 * count
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import sys
                print(f"[ba] all-reduce hook failed: {e!r}", file=sys.stderr)
                return 1

        self._cb = _lib.ALLREDUCE_CB(_hook)  # keep alive
        _lib.check(_lib.lib().ba_lm_set_comm(nlp.handle, self.rank, self.world, C.c_void_p(self.buf.data_ptr()),
                                             n.value, self._cb, None))


def allreduce_sum_numpy(a, group=None):
    """Sum a numpy float64 array over the ranks (gloo on CPU); used by the CPU tests of the sharding logic."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()
