"""Multi-GPU sharding of the photon walk: one process per GPU, independent
photon-id ranges, one sum-reduce of the voxel grid at the end.

Photons are independent (as pixels are in the reference, path_tracing_fix1.py:
144-148) and every photon's RNG stream is a function of (seed, photon id) only,
so rank r of P simply traces ids [r*N/P, (r+1)*N/P) into a private grid; there is
no data-path collective until the final reduce (RCCL over xGMI on GPUs, gloo in
the CPU tests).  With the u64 fixed-point tally the reduced grid is bit-identical
for every P.
"""
import numpy as np

import torch
import torch.distributed as dist


def shard_range(n_photons, rank, world_size):
    """Contiguous, exhaustive, non-overlapping id range of ``rank``: (offset, count)."""
    lo = n_photons * rank // world_size
    hi = n_photons * (rank + 1) // world_size
    return lo, hi - lo


class _DevMem:
    """Adapter exposing a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr, False), version=2, strides=None)


_TYPESTR = {0: "<f4", 1: "<f8", 2: "<i8"}  # u64 fixed point reduced as int64 (same bits, same sum mod 2^64)


def device_grid_tensor(ctx):
    ptr, nbytes = ctx.grid_device_ptr()
    ts = _TYPESTR[ctx._tally]
    n = nbytes // int(ts[2])
    return torch.as_tensor(_DevMem(ptr, n, ts), device="cuda:%d" % ctx.device_id)


def device_counter_tensors(ctx):
    ptr, nbytes = ctx.counters_device_ptr()
    ints = torch.as_tensor(_DevMem(ptr, 2, "<i8"), device="cuda:%d" % ctx.device_id)
    flts = torch.as_tensor(_DevMem(ptr + 16, 8, "<f8"), device="cuda:%d" % ctx.device_id)
    return ints, flts


def reduce_device(ctx, group=None, dst=None, wait=True):
    """Sum-reduce grid and counters in place in HBM with torch.distributed (backend "nccl" == RCCL on ROCm),
    dst=None: all-reduce.  The collectives are issued with the ctx's own HIP stream as torch's current stream, so
    they are ordered behind the launch that produced the grid and in front of whatever the caller enqueues ON THE CTX
    next (lt_read_grid, counters readback, the next zero_tally): torch's NCCL process group makes its communication
    stream wait for the current stream's work and, at ``Work.wait()``, the current stream for the communication --
    both are stream-level events.

    wait=True (default): the host then blocks until the ctx stream has drained, so the reduced grid may be consumed
    from ANY stream afterwards (``device_grid_tensor(ctx).cpu()`` on torch's default stream included).
    wait=False: returns without a host or torch-stream sync -- the reduced grid may then only be consumed THROUGH THE
    CTX (its stream: read_grid / read_counters / lt_stream) or after ``ctx.sync()``; a torch-side read on another
    stream is NOT ordered behind the reduce.  bench.py uses this form so that, with several jobs in flight, one job's
    reduce travels over xGMI while another job walks."""
    stream = torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda", ctx.device_id))
    tensors = (device_grid_tensor(ctx),) + device_counter_tensors(ctx)
    with torch.cuda.stream(stream):
        works = []
        for t in tensors:
            if dst is None:
                works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
            else:
                works.append(dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()          # stream-level for the NCCL backend: the ctx stream waits, the host does not
    if wait:
        ctx.sync()


def reduce_host(grid, counters, group=None, dst=None):
    """Same reduction on host arrays (gloo); used by the CPU tests of the N>1
    path and as a fallback transport when no GPU collective is wanted.
    ``grid``: ndarray (float32/float64/uint64); ``counters``: dict as read_counters()."""
    g = torch.from_numpy(np.ascontiguousarray(grid).view(np.int64) if grid.dtype == np.uint64 else np.ascontiguousarray(grid))
    keys_i = ("photons", "steps")
    keys_f = ("w_absorbed", "w_lost_outside_grid", "w_escaped_top", "w_escaped_bottom", "w_escaped_mesh",
              "w_specular", "w_roulette_net", "w_capped")
    ci = torch.tensor([int(counters[k]) for k in keys_i], dtype=torch.int64)
    cf = torch.tensor([float(counters[k]) for k in keys_f], dtype=torch.float64)
    for t in (g, ci, cf):
        if dst is None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    out = g.numpy()
    if grid.dtype == np.uint64:
        out = out.view(np.uint64)
    c = {k: int(v) for k, v in zip(keys_i, ci.tolist())}
    c.update({k: float(v) for k, v in zip(keys_f, cf.tolist())})
    return out.reshape(grid.shape), c
