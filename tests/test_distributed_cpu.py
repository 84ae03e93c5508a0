"""CPU suite: the N > 1 path.  world_size-2 gloo processes shard the photon ids
with shard_range, walk their range (here with the CPU oracle standing in for the
device walk -- this is a test of the sharding + reduction logic, not of the
kernels) and sum-reduce grid + counters with reduce_host.  The reduced u64
fixed-point grid must be BIT-IDENTICAL to a single-process run."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_photons, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from light_transport_amd.distributed import reduce_host, shard_range
    from tests import scenes as S
    off, cnt = shard_range(n_photons, rank, world)
    _, fx, c = S.two_layer(n=24).oracle().run(cnt, seed=21, photon_offset=off, want_fx=True, want_f64=False)
    red, cr = reduce_host(fx, c, dst=None)
    g64, _, c64 = S.two_layer(n=24).oracle().run(cnt, seed=21, photon_offset=off)
    red64, _ = reduce_host(g64, c64, dst=0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "reduced.npz"), fx=red, f64=red64, photons=cr["photons"], steps=cr["steps"],
                 absorbed=cr["w_absorbed"])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from light_transport_amd.distributed import shard_range
    for n in (0, 1, 7, 10 ** 7, 10 ** 8 + 3):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (o0, c0), (o1, _) in zip(parts, parts[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_two_rank_gloo_reduce_matches_single_run(tmp_path):
    sys.path.insert(0, ROOT)
    from tests import scenes as S
    n = 3001
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "reduced.npz"))
    g64, fx, c = S.two_layer(n=24).oracle().run(n, seed=21, want_fx=True)
    assert np.array_equal(got["fx"], fx)                       # bit-identical for every P
    np.testing.assert_allclose(got["f64"], g64, rtol=1e-12, atol=1e-13)
    assert int(got["photons"]) == n and int(got["steps"]) == c["steps"]
    assert abs(float(got["absorbed"]) - c["w_absorbed"]) < 1e-9
