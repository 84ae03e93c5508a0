"""GPU suite: the N > 1 path with REAL device walks.  Two processes (gloo rendezvous on 127.0.0.1) share the box's one
GPU, each traces its shard_range of the photon ids through the C ABI into its own context, and the grids + counters
are sum-reduced (reduce_host; RCCL refuses two ranks on one device, so the device-side reduce_device path is
exercised by bench.py under torch.distributed.run instead).  The reduced fixed-point grid must be bit-identical to a
single-process run of the whole id range -- photon streams depend on (seed, id) only."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_photons, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import light_transport_amd as lt
    from light_transport_amd.distributed import reduce_host, shard_range
    from tests import scenes as S
    ctx = lt.Context(0)
    prob = S.two_layer(n=64)
    prob.apply(ctx, "u64fx")
    off, cnt = shard_range(n_photons, rank, world)
    ctx.launch(cnt, seed=77, photon_offset=off)
    ctx.sync()
    red, c = reduce_host(ctx.read_grid_raw(), ctx.read_counters(), dst=None)
    if rank == 0:
        ctx.zero_tally()
        ctx.launch(n_photons, seed=77)
        ctx.sync()
        whole, cw = ctx.read_grid_raw(), ctx.read_counters()
        np.savez(os.path.join(out_dir, "r.npz"), equal=np.array_equal(red, whole), steps=c["steps"], steps_whole=cw["steps"],
                 photons=c["photons"], absorbed=c["w_absorbed"], absorbed_whole=cw["w_absorbed"], total=int(whole.sum() > 0))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_device_match_single_run(tmp_path):
    n = 200001
    port = 29600 + (os.getpid() % 1500)
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    r = np.load(os.path.join(str(tmp_path), "r.npz"))
    assert bool(r["equal"]) and int(r["total"]) == 1
    assert int(r["steps"]) == int(r["steps_whole"]) and int(r["photons"]) == n
    assert abs(float(r["absorbed"]) - float(r["absorbed_whole"])) < 1e-6


def test_lt_reduce_grid_with_a_one_rank_rccl_communicator(ctx):
    """The C-host form of the reduce: lt_reduce_grid(ctx, ncclComm_t, root) on a communicator created directly from
    librccl (1 rank: the sum is the identity, which checks symbol loading, datatypes and stream use end to end)."""
    import ctypes as C
    import glob
    import torch
    import light_transport_amd as lt
    sys.path.insert(0, ROOT)
    from tests import scenes as S
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["librccl.so", "librccl.so.1"]
    rccl = None
    for c in cands:
        try:
            rccl = C.CDLL(c, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("librccl.so not loadable")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        prob = S.two_layer(n=32)
        for dtype in ("u64fx", "f64", "f32"):
            prob.apply(ctx, dtype)
            ctx.launch(20000, seed=3, f32_walk=dtype == "f32"); ctx.sync()
            before, cb = ctx.read_grid_raw(), ctx.read_counters()
            for root in (-1, 0):
                rc = lt.lib().lt_reduce_grid(ctx._h, comm, C.c_int(root))
                assert rc == 0, lt.lib().lt_last_error(ctx._h)
                ctx.sync()
                assert np.array_equal(ctx.read_grid_raw(), before)
                ca = ctx.read_counters()
                assert ca["steps"] == cb["steps"] and ca["w_absorbed"] == cb["w_absorbed"]
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
