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


@pytest.mark.parametrize("flags", [["--workload", "c2", "--steps", "6"], ["--workload", "c5", "--steps", "2"]])
def test_bench_self_launched_world_1_runs_reduce_device_on_rccl(flags):
    """bench.py's own launcher (what `python bench.py --gpus N` does for N > 1) at world size 1: a fresh child under
    torch.distributed.run, process group "nccl" (= RCCL), every job's grid + counters reduced by
    distributed.reduce_device on the job's stream.  One JSON line, the regime probe ran, the reduces ran on RCCL."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    code = "import sys, bench; sys.exit(bench.self_launch(1, sys.argv[1:]))"
    cmd = [sys.executable, "-c", code, "--gpus", "1", "--warmup", "1", "--no-cpu-baseline"] + flags
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    K = int(flags[-1])
    assert out["n_gpus"] == 1 and out["steps"] == K and out["value"] > 1e10
    probe = out["config"]["regime_probe"]
    assert probe and probe["chosen"] == out["config"]["regime"] and "one_call_ms" in probe
    red = out["config"]["reduce"]
    assert red["backend"] == "nccl" and red["calls_rank0"] >= K + 1       # timed + warm-up (+ the probe's jobs)
    assert out["roofline"]["frac"] > 0.02


def test_bench_under_the_launcher_keeps_the_plain_rate():
    """The same regime, the same K, plain (`python bench.py`) and under bench.py's own launcher at world size 1 (torch.distributed.run,
    RCCL process group, every job's grid reduced on its stream), interleaved on this box: the launched run must reach >= 95 % of the
    plain one.  Round 2 measured 73.8e9 against 77-80e9 once and nobody looked again; the line also has to say that the jobs in
    flight really overlapped on this rank (config.overlap_factor_*, serialised_ranks)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    flags = ["--gpus", "1", "--inflight", "3", "--steps", "12", "--warmup", "3", "--no-alone", "--no-cpu-baseline", "--extras", "none"]

    def run(launched):
        cmd = ([sys.executable, "-c", "import sys, bench; sys.exit(bench.self_launch(1, sys.argv[1:]))"] if launched
               else [sys.executable, os.path.join(ROOT, "bench.py")]) + flags
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    plain, launched = [], []
    for _ in range(2):
        plain.append(run(False)); launched.append(run(True))
    best_plain, best_launched = max(d["value"] for d in plain), max(d["value"] for d in launched)
    for d in plain + launched:
        assert d["config"]["regime"] == "three_jobs" and d["config"]["serialised_ranks"] == 0, d["config"]
        assert d["config"]["overlap_factor_min"] > 1.8, d["config"]      # three jobs in flight really were in flight together
    assert launched[0]["config"]["reduce"]["backend"] == "nccl" and launched[0]["config"]["reduce"]["calls_rank0"] >= 15
    assert best_launched >= 0.95 * best_plain, (best_plain, best_launched)


def _reduce_worker(rank, world, port, n_photons, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import light_transport_amd as lt
    from light_transport_amd.distributed import device_grid_tensor, reduce_device, shard_range
    from tests import scenes as S
    ctx = lt.Context(rank)
    S.two_layer(n=64).apply(ctx, "u64fx")
    off, cnt = shard_range(n_photons, rank, world)
    ctx.launch(cnt, seed=77, photon_offset=off)
    ctx.sync()
    np.save(os.path.join(out_dir, "own%d.npy" % rank), ctx.read_grid_raw())
    steps_own = ctx.read_counters()["steps"]
    np.save(os.path.join(out_dir, "steps%d.npy" % rank), np.array([steps_own]))
    reduce_device(ctx, dst=None)                       # default wait=True: consumable from any stream afterwards
    red_torch = device_grid_tensor(ctx).cpu().numpy().view(np.uint64)     # torch's default stream
    np.save(os.path.join(out_dir, "red%d.npy" % rank), ctx.read_grid_raw())
    np.save(os.path.join(out_dir, "redt%d.npy" % rank), red_torch)
    np.save(os.path.join(out_dir, "rsteps%d.npy" % rank), np.array([ctx.read_counters()["steps"]]))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_device_two_ranks_rccl(tmp_path):
    """The N > 1 device path proper (runs where the box has >= 2 GPUs): two ranks, one GPU each, RCCL all-reduce through
    distributed.reduce_device.  The reduced fixed-point grid must equal the sum of the per-rank grids bit for bit, on the
    ctx stream AND -- with the default wait=True -- when read through torch's default stream."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); world-size-1 RCCL and two-process gloo cover the rest")
    n, d = 100001, str(tmp_path)
    port = 29650 + (os.getpid() % 1500)
    mp.spawn(_reduce_worker, args=(2, port, n, d), nprocs=2, join=True)
    own = [np.load(os.path.join(d, "own%d.npy" % r)) for r in range(2)]
    want = own[0] + own[1]
    assert want.sum() > 0
    steps = sum(int(np.load(os.path.join(d, "steps%d.npy" % r))[0]) for r in range(2))
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(d, "red%d.npy" % r)), want)
        assert np.array_equal(np.load(os.path.join(d, "redt%d.npy" % r)).reshape(want.shape), want)
        assert int(np.load(os.path.join(d, "rsteps%d.npy" % r))[0]) == steps
