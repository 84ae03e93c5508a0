#!/usr/bin/env python3
"""Headline benchmark of the photon-transport hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (N > 1 without RANK in the environment: starts the line above as a child process)

One "step" = one complete job on every rank: zero the tally, trace the workload's photons (f64 walk, XORWOW, f64 tally;
deposits go through the log-structured tally: walk -> deposit log -> tile partition -> LDS reduce) and -- for N > 1 --
sum-reduce the voxel grid + counters to rank 0 with RCCL, enqueued on the job's own stream.  Ranks trace disjoint
photon-id ranges (weak scaling); there is no other collective.  Inputs are synthetic by nature (the scene is ~100 bytes
of constants, resident in HBM before the timed region).

Workloads (BASELINE.json configs):
  c2 (default)  1e7 photons per GPU, homogeneous semi-infinite slab, 256^3 grid -- the config the metric is quoted on
  c3            1e7 photons, two-layer skin model, 256^3 grid (0.05 mm)
  c4            1e7 photons, Cornell cavity + cone (30 triangles, BVH), 256^3 grid, cosine source on the ceiling quad
  c5            1.25e7 photons per GPU (the per-GPU share of 1e8 over 8), two-layer skin model, 512^3 grid (1 GiB f64)
The default run (c2, N = 1) measures c3, c4 and c5 AFTER the headline's timed region as well and reports them as flat
roofline.cN_* keys (--extras): jobs in flight, one launch alone, its kernels, the CPU oracle beside each.

Regimes (how a rank keeps its GPU busy; results are identical):
  one_call      ONE context; every lt_launch is cut into sub-batches that alternate between the context's two lanes, so
                that one batch's log reduction runs beside the next batch's walk (lt_set_overlap 2)      [--inflight 1]
  two_jobs      TWO contexts take the steps in turn (two complete jobs in flight)                         [--inflight 2]
  three_jobs    THREE contexts (256^3 workloads: the logs of three jobs must fit)                         [--inflight 3]
  walk_train    jobs in flight whose walks run one after another at 3/4 of the resident workgroups; the free quarter of
                the register file carries the previous job's log reduction                               [--inflight 4]
  one_at_a_time ONE context, one lane: kernels strictly back to back                        [--inflight 1 --overlap 1]
By default a short untimed probe runs all of them and the timed region uses the fastest; the probe's numbers are reported.

Prints ONE JSON line on rank 0:  metric = photon-steps/s over all ranks, plus
  roofline      algorithmic tally bytes (16 B per photon-step for the f64 tally: 8 B read + 8 B write of one voxel)
                per launch / the time the device takes per launch, against the 8 TB/s HBM peak; per-kernel durations of
                one job alone on the device (HIP events on the ctx stream) with their own fractions; measured HBM
                traffic (rocprofv3 PMC passes, profiles/traffic.json -- only if taken on these very kernel sources);
  readback      D2H of the grid into pinned host memory, and the readback-inclusive rate;
  cpu_baseline  the CPU oracle (oracle/, a port -- the reference has no such path) on the host cores on a bounded
                sample of the same workload.
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Jobs / lanes in flight need their streams on DIFFERENT hardware queues: the HIP runtime multiplexes a process's
# streams onto GPU_MAX_HW_QUEUES (default 4) queues, and once torch + RCCL have taken theirs two streams can end up
# sharing one, which serialises their kernels (measured: 57 instead of 39 ms per step under torch.distributed.run).
# Must be set before the runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

INF = float("inf")
# BASELINE.json configs (sizes and media: SURVEY.md 8(d)).  scene = "layers" (z_bounds, medium_idx) or "cornell" (the
# reference's Cornell cube + cone, nb:LTS#11-16: 30 triangles, BVH, cosine source on the ceiling light quad).
WORKLOADS = {
    "c2": dict(photons=10 ** 7, grid=256, voxel=0.1, media=[(0.1, 10.0, 0.9, 1.0)], z_bounds=[0.0, INF], medium_idx=[0],
               text="C2: %.0e photons per GPU, homogeneous semi-infinite slab (mu_a=0.1, mu_s=10, g=0.9, n=1), %d^3 voxel grid "
                    "(%.3g mm), pencil beam"),
    "c3": dict(photons=10 ** 7, grid=256, voxel=0.05, media=[(0.43, 10.7, 0.79, 1.5), (0.27, 18.7, 0.82, 1.4)],
               z_bounds=[0.0, 0.1, INF], medium_idx=[0, 1],
               text="C3: %.0e photons per GPU, two-layer skin model (epidermis 0.1 mm n=1.5 / dermis n=1.4, air above), %d^3 voxel "
                    "grid (%.3g mm), pencil beam"),
    "c4": dict(photons=10 ** 7, grid=256, voxel=15.0 / 256, media=[(0.1, 10.0, 0.9, 1.0), (1.0, 5.0, 0.8, 1.5)], cornell=7.5,
               text="C4: %.0e photons per GPU, Cornell cavity (half-width 7.5, C2's medium) + 10-triangle cone of a second medium "
                    "(n=1.5): 30 triangles + BVH, %d^3 voxel grid (%.3g), cosine source on the 2x2 ceiling quad"),
    "c5": dict(photons=12500000, grid=512, voxel=0.025, media=[(0.43, 10.7, 0.79, 1.5), (0.27, 18.7, 0.82, 1.4)],
               z_bounds=[0.0, 0.1, INF], medium_idx=[0, 1],
               text="C5 per-GPU share: %.3g photons per GPU (1e8 over 8 GPUs), two-layer skin model (epidermis 0.1 mm n=1.5 / "
                    "dermis n=1.4, air above), %d^3 voxel grid (%.3g mm), pencil beam"),
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_STEP = {"f32": 8, "f64": 16, "u64fx": 16}
REC_VALUE_BYTES = {"f32": 4, "f64": 8, "u64fx": 8}
_MESH = {}


def cornell_mesh(dim):
    """Config 4's geometry from the package's own mirror of the reference's scene builders (src/cornell_box.py,
    src/bvh_new.py): walls face the cavity (front = medium 0, back = exterior), the cone's faces point outwards."""
    if dim not in _MESH:
        from light_transport_amd.src import bvh_new as B, constants as K, cornell_box as cb
        walls = (cb.get_cornell_box(dim, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(dim, K.GLASS_MAT)
                 + cb.get_light_quad(dim, K.GLASS_MAT))
        cone = cb.get_cone(K.GLASS_MAT)
        for t in walls:
            t.med_front, t.med_back = 0, -1
        for t in cone:
            t.med_front, t.med_back = 0, 1
        ordered, linear = B.build_linear_bvh(walls + cone, 1)
        _MESH[dim] = dict(verts=B.triangles_array(ordered), med_front=np.array([t.med_front for t in ordered], np.int32),
                          med_back=np.array([t.med_back for t in ordered], np.int32), nodes=B.linear_bvh_arrays(linear))
    return _MESH[dim]


def grid_origin(wl):
    half = wl["grid"] * wl["voxel"] / 2
    return (-half, -half, -half) if "cornell" in wl else (-half, -half, 0.0)


def source_of(wl):
    if "cornell" in wl:
        d = wl["cornell"]
        return dict(type=1, pos=(-1.0, d, -1.0), dir=(0.0, -1.0, 0.0), extra=(2.0, 0.0, 0.0, 0.0, 0.0, 2.0), start_medium=0)
    return dict(type=0, pos=(0.0, 0.0, 0.0), dir=(0.0, 0.0, 1.0), extra=None, start_medium=0)


def configure(ctx, wl, tally):
    ctx.set_media(wl["media"])
    if "cornell" in wl:
        m = cornell_mesh(wl["cornell"])
        ctx.set_mesh(m["verts"], m["med_front"], m["med_back"], m["nodes"])
    else:
        ctx.set_layers(wl["z_bounds"], wl["medium_idx"], 1.0, 1.0)
    ctx.set_grid((wl["grid"],) * 3, grid_origin(wl), (wl["voxel"],) * 3, tally)
    s_ = source_of(wl)
    ctx.set_source(s_["type"], s_["pos"], s_["dir"], s_["extra"], s_["start_medium"])


def kernel_sources_sha():
    """Identity of the kernel sources a PMC pass was taken on (profiles/traffic.json carries the same hash)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "light_transport_amd", "csrc", "*.[hic]*")) + [os.path.join(ROOT, "include", "lt.h")]):
        if f.endswith(".o"):
            continue
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def effective_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants a 16-CPU quota; oversubscribing it makes the oracle slower, not faster)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(np.ceil(q / p_))))
        except Exception:
            pass
    return n


def cpu_baseline(wl, wl_name, target_seconds=12.0):
    """CPU oracle (port) on every host core, bounded sample of the workload."""
    from oracle import oracle as O
    cores = effective_cores()
    if "cornell" in wl:
        src = source_of(wl)
        sc = O.OracleScene(wl["media"], (wl["grid"],) * 3, grid_origin(wl), (wl["voxel"],) * 3, mesh=cornell_mesh(wl["cornell"]),
                           source=dict(src, extra=src["extra"] or (0.0,) * 6))
    else:
        sc = O.OracleScene(wl["media"], (wl["grid"],) * 3, grid_origin(wl), (wl["voxel"],) * 3,
                           layers=dict(z_bounds=wl["z_bounds"], medium_idx=wl["medium_idx"]))
    t0 = time.perf_counter()
    _, _, c = sc.run(50000, seed=0, threads=cores)
    probe = time.perf_counter() - t0
    rate, per_photon = c["steps"] / probe, c["steps"] / 50000.0
    n = int(min(wl["photons"], max(100000, target_seconds * rate / per_photon)))
    t0 = time.perf_counter()
    _, _, c = sc.run(n, seed=0, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": c["steps"] / dt, "unit": "photon-steps/s", "cores": cores, "kind": "port",
            "sample": "first %d photons of the same %s workload (f64, %d^3 f64 grid), %.1f s wall, pthreads" % (
                n, wl_name.upper(), wl["grid"], dt),
            "photons_per_sec": n / dt}


class Device:
    """torch plumbing of a real run: device selection, synchronisation, scalars for the rank agreement."""
    name = "cuda"

    def __init__(self, local_rank):
        import torch
        self.torch = torch
        torch.cuda.set_device(local_rank)

    def sync(self):
        self.torch.cuda.synchronize()

    def scalar(self, v, dtype):
        return self.torch.tensor([v], dtype=dtype, device="cuda")

    def host_buffer(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, pin_memory=True).numpy()


class HostDevice(Device):
    """Stand-in used by the CPU test of the N > 1 control flow (gloo, recording contexts): no GPU is touched."""
    name = "cpu"

    def __init__(self, local_rank):
        import torch
        self.torch = torch

    def sync(self):
        pass

    def scalar(self, v, dtype):
        return self.torch.tensor([v], dtype=dtype)

    def host_buffer(self, nbytes):
        return np.empty(nbytes, dtype=np.uint8)


def self_launch(n, argv):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <argv>` on a free local port
    and return its exit code (stdout / stderr are inherited, so rank 0's JSON line is this command's JSON line)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = str(s_.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL / dmabuf IPC on this pool's host driver
    return subprocess.call(cmd, env=env)


def measure_extra(name, make_ctx, args, cpu_seconds):
    """One more BASELINE config after the headline's timed region (single GPU): flat keys for the JSON line.
    (a) throughput: two or three contexts take whole jobs in turn (the two_jobs / walk_train / three_jobs regimes of the
        headline, each measured; the fastest is the figure);
    (b) latency: ONE lt_launch alone, library defaults (what a trace_photons caller gets);
    (c) its kernels on one lane, nothing beside them (tail split off: the per-kernel figures);
    (d) the CPU oracle on the same workload, bounded sample."""
    wl = WORKLOADS[name]
    n = wl["photons"]
    half = 3 if args.f32_walk else 2
    cs = []
    try:
        def add_ctx():
            c = make_ctx()
            configure(c, wl, args.tally)
            c.set_tally_mode(args.tally_mode); c.set_overlap(1); c.set_launch_config(half, 256)
            if args.tally_mode != "atomic":
                c.reserve_log(n)
            cs.append(c)
        for _ in range(2):
            add_ctx()

        def jobs(k, seed0):
            d, steps = len(cs), 0
            for c in cs:
                c.sync()
            t0 = time.perf_counter()
            for j in range(k):
                c = cs[j % d]
                if j >= d:
                    c.sync(); steps += c.read_counters()["steps"]
                c.zero_tally(); c.launch(n, seed=seed0 + j, f32_walk=args.f32_walk)
            for j in range(max(0, k - d), k):
                c = cs[j % d]
                c.sync(); steps += c.read_counters()["steps"]
            return (time.perf_counter() - t0) / k * 1e3, steps / k
        # ways of keeping jobs in flight (bench.py's two_jobs, walk_train at depth 2 and -- where three deposit logs fit, the
        # 256^3 grids -- three_jobs); the fastest is reported, all are kept
        jobs(4, 700)                      # pilot batch of the scene, log sizing, warm-up
        ms, steps = jobs(8, 710)
        regime, both = "two_jobs", {name + "_two_jobs_ms": ms}
        if not args.f32_walk:
            for c in cs:
                c.set_tuning("serial_walks", 1); c.set_launch_config(3, 256)
            jobs(4, 740)
            ms_t, steps_t = jobs(8, 750)
            both[name + "_walk_train_ms"] = ms_t
            if ms_t < ms:
                ms, steps, regime = ms_t, steps_t, "walk_train"
            for c in cs:
                c.set_tuning("serial_walks", -1); c.set_launch_config(half, 256)
        if wl["grid"] <= 256:
            add_ctx()
            jobs(6, 760)
            ms_3, steps_3 = jobs(18, 770)
            both[name + "_three_jobs_ms"] = ms_3
            if ms_3 < ms:
                ms, steps, regime = ms_3, steps_3, "three_jobs"
            cs.pop().close()
        out = {name + "_ms": ms, name + "_steps_per_s": steps / (ms * 1e-3), name + "_photons_per_s": n / (ms * 1e-3),
               name + "_frac": steps * BYTES_PER_STEP[args.tally] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, name + "_regime": regime,
               name + "_steps_per_job": steps}
        out.update(both)
        cs.pop().close()
        c = cs[0]
        c.set_launch_config(0, 0); c.set_overlap(0)
        t_ = []
        for k in range(6):                # overlap auto settles over the first launches (one lane / two lanes, keeps the faster)
            c.zero_tally(); c.launch(n, seed=720 + k, f32_walk=args.f32_walk); c.sync()
            t_.append(c.last_kernel_ms())
        info = c.last_log_info()
        out[name + "_one_launch_ms"] = float(np.mean(t_[3:]))
        out[name + "_one_launch_steps_per_s"] = steps / (out[name + "_one_launch_ms"] * 1e-3)
        out[name + "_one_launch_lanes"] = info["lanes"] if info else 1
        c.set_overlap(1); c.set_tuning("tail_split", 0)
        st, t1 = None, []
        for k in range(3):
            c.zero_tally(); c.launch(n, seed=730 + k, f32_walk=args.f32_walk); c.sync()
            t1.append(c.last_kernel_ms()); st = c.last_log_stages()
        c.set_tuning("tail_split", -1)
        out[name + "_alone_job_ms"] = float(np.mean(t1[1:]))
        if st:
            rec, rb_ = st["records"], 4 + REC_VALUE_BYTES[args.tally]
            for k_ in ("walk_ms", "scan_ms", "partition_ms", "reduce_ms"):
                out["%s_alone_%s" % (name, k_)] = st[k_]
            out[name + "_records_per_job"] = rec
            two_pass = wl["grid"] ** 3 > 1024 * 16384
            p_ = st["partition_ms"] + (st["scan_ms"] if two_pass else 0.0)
            if p_ > 0 and st["reduce_ms"] > 0:
                out[name + "_partition_frac"] = rec * (2 * rb_ - 2) / (p_ * 1e-3) / 1e9 / HBM_PEAK_GBS
                out[name + "_reduce_frac"] = rec * (rb_ - 2) / (st["reduce_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out[name + "_walk_steps_per_s"] = steps / (st["walk_ms"] * 1e-3)
    finally:
        for c in cs:
            c.close()
    if cpu_seconds:
        cb = cpu_baseline(wl, name, target_seconds=cpu_seconds)
        out[name + "_cpu_steps_per_s"], out[name + "_cpu_cores"], out[name + "_cpu_sample"] = cb["value"], cb["cores"], cb["sample"]
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough that the edges of the timed region -- the pipeline of jobs in flight filling, and the last
    # jobs draining before the closing barrier -- weigh little (K = 10: 36.1-37.7 ms per C2 job, K = 40: 34.6-34.7);
    # the whole run still takes well under a minute
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--photons", type=int, default=0, help="photons per GPU per step (default: the workload's)")
    ap.add_argument("--tally", default="f64", choices=["f32", "f64", "u64fx"])
    ap.add_argument("--f32-walk", action="store_true", help="f32 walk arithmetic (default f64, the reference's dtype)")
    ap.add_argument("--tally-mode", default="log", choices=["log", "atomic", "auto"],
                    help="log: deposit log + tile partition + LDS reduce (default); atomic: one global atomic per deposit")
    ap.add_argument("--inflight", type=int, default=0,
                    help="contexts taking the steps in turn: 2 / 3 = that many jobs in flight, 4 = the walk train (jobs in flight, walks "
                         "one after another at 3/4 occupancy), 1 = one context; 0 (default) = an untimed probe picks the fastest regime")
    ap.add_argument("--overlap", type=int, default=-1,
                    help="lanes inside one launch (lt_set_overlap) for --inflight 1: 2 = one_call, 1 = one_at_a_time")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the single-job reference launches and the readback after the timed region (keeps a rocprofv3 "
                         "kernel trace of this command to launches of the timed regime only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", default="auto",
                    help="BASELINE configs measured AFTER the timed region (N = 1 only; each: jobs in flight, one launch alone, "
                         "its kernels, the CPU oracle) and reported as flat roofline.cN_* keys: 'auto' = c3,c4,c5 for the default "
                         "c2 run, 'none', or a comma-separated list")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)        # tests: gloo
    ap.add_argument("--ctx-factory", default="", help=argparse.SUPPRESS)        # tests: module:callable returning a recording ctx
    args = ap.parse_args(argv)
    wl = WORKLOADS[args.workload]
    per_gpu = args.photons or wl["photons"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "RANK" in os.environ
    if args.gpus != world and distributed:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and not distributed:
        # `python bench.py --gpus N` from a plain shell: become the launcher.  One rank per GPU is started through
        # torch.distributed.run as a FRESH child process -- nothing in this process has touched the device (torch is not
        # even imported yet) -- and this process only relays the child's exit code.  (Role of Numba's thread pool
        # spreading render_scene's rows over the cores, S/path_tracing_fix1.py:139-148.)
        sys.exit(self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    fake = bool(args.ctx_factory)
    dev = HostDevice(local_rank) if fake else Device(local_rank)
    torch = dev.torch
    dist = None
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    n_reduces = [0]
    if fake:
        mod, attr = args.ctx_factory.split(":")
        make_ctx = getattr(importlib.import_module(mod), attr)
        reduce_one = lambda c: c.reduce_to(dist, 0)                              # noqa: E731
    else:
        import light_transport_amd as lt
        from light_transport_amd import distributed as ltd
        make_ctx = lambda: lt.Context(local_rank)                                # noqa: E731
        # RCCL sum of grid + counters to rank 0, enqueued on the ctx stream; the host does not wait (finish() syncs the ctx)
        reduce_one = lambda c: ltd.reduce_device(c, dst=0, wait=False)           # noqa: E731

    def reduce_ctx(c):
        n_reduces[0] += 1
        reduce_one(c)

    offset = rank * per_gpu     # disjoint id ranges; streams depend on (seed, id) only

    # ---- contexts: A serves one_call / one_at_a_time, A + B serve two_jobs
    def new_ctx():
        c = make_ctx()
        configure(c, wl, args.tally)
        c.set_tally_mode(args.tally_mode)
        return c

    pool = [new_ctx()]           # contexts, created when a regime first needs them
    # name -> (contexts taking the jobs in turn, lanes per launch).  three_jobs: while one job reduces, TWO walks keep all
    # four waves per SIMD busy (C2: 35.8 against 37.0 ms with two jobs); its three deposit logs only fit beside the
    # other regimes' buffers for the 256^3 workload
    REGIMES = {"one_call": (1, 2), "two_jobs": (2, 1), "one_at_a_time": (1, 1)}
    if wl["grid"] <= 256:
        REGIMES["three_jobs"] = (3, 1)
    # walk_train: jobs in flight whose WALKS run one after another (lt_set_tuning "serial_walks"), each at 3 of the 4 resident
    # workgroups per CU (f32: 4 of 5) while the free quarter of every SIMD's register file carries the previous job's partition
    # and tile reduce.  Measured behind three_jobs on every workload by the end of round 4 (DESIGN.md "Overlap": a walk loses
    # speed in proportion to the residency it gives up); kept in the probe as a measured alternative.
    if not args.f32_walk:
        REGIMES["walk_train"] = (int(os.environ.get("LT_BENCH_TRAIN_DEPTH", 3 if wl["grid"] <= 256 else 2)), 1)

    def apply_regime(name):
        depth, lanes = REGIMES[name]
        while len(pool) < depth:
            pool.append(new_ctx())
        cs = pool[:depth]
        for c in cs:
            c.set_overlap(lanes)
            if hasattr(c, "set_tuning"):
                c.set_tuning("serial_walks", 1 if name == "walk_train" else -1)
            # jobs in flight: each job's walk takes half of the resident workgroups (f64: 2 of 4 per CU; f32: 3 of 5);
            # the walk train: three of four
            train_bpc = int(os.environ.get("LT_BENCH_TRAIN_BPC", "3"))       # (measurement override)
            c.set_launch_config((train_bpc if name == "walk_train" else (3 if args.f32_walk else 2)) if depth >= 2 else 0, 256 if depth >= 2 else 0)
            if args.tally_mode != "atomic":
                c.reserve_log(per_gpu)    # scratch allocation is set-up, not part of a step (matters when --warmup 0)
        return cs

    every = {"walk_ms": 0.0, "partition_ms": 0.0, "reduce_ms": 0.0, "batches": 0, "launches": 0}

    def note(c):
        """Stage sums of the launch that has just completed on c (HIP events on its streams): EVERY launch of this
        process is counted -- probe, warm-up, timed and reference launches -- so that the per-dispatch averages can be
        held against a rocprofv3 --kernel-trace --stats summary of the same command."""
        st = c.last_log_stages()
        if st is not None:
            for k in ("walk_ms", "partition_ms", "reduce_ms", "batches"):
                every[k] += st[k]
            every["launches"] += 1
        return st

    def run_jobs(cs, n, seed0):
        """n complete jobs over the contexts cs in turn -- for N > 1 INCLUDING each job's reduce to rank 0, exactly as the
        timed region runs them (the regimes differ in how many streams they keep busy beside RCCL's) -- host-timed: ms
        per job.  Every rank runs the same sequence, so the collectives pair up."""
        for c in cs:
            c.sync()
        t0 = time.perf_counter()
        for k in range(n):
            c = cs[k % len(cs)]
            if k >= len(cs):
                if distributed:
                    reduce_ctx(c)
                c.sync(); note(c)
            c.zero_tally(); c.launch(per_gpu, seed=seed0 + k, photon_offset=offset, f32_walk=args.f32_walk)
        for k in range(max(0, n - len(cs)), n):
            c = cs[k % len(cs)]
            if distributed:
                reduce_ctx(c)
            c.sync(); note(c)
        return (time.perf_counter() - t0) / n * 1e3

    probe = None
    if args.inflight == 4:
        regime = "walk_train"
    elif args.inflight == 3:
        regime = "three_jobs"
    elif args.inflight == 2:
        regime = "two_jobs"
    elif args.inflight == 1:
        regime = "one_at_a_time" if args.overlap == 1 else "one_call"
    elif args.tally_mode == "atomic":
        regime = "one_at_a_time"
    else:
        # Untimed probe: which regime keeps THIS device busiest?  (Two streams only overlap when the runtime gives them
        # separate hardware queues -- see GPU_MAX_HW_QUEUES above -- so it is measured, not assumed.)
        probe = {}
        for name in ("one_at_a_time", "one_call", "two_jobs", "three_jobs", "walk_train"):
            if name not in REGIMES:
                continue
            cs = apply_regime(name)
            run_jobs(cs, len(cs), 900)                    # sizes the logs, pilot batch of a new scene
            # jobs in flight: 24 jobs (0.9 s), so that the pipeline's fill and drain -- a third of a 6-job probe with three
            # jobs in flight -- weigh under 2 % and regimes a millisecond apart are told apart; the one-context regimes
            # (never within 5 % of the winners) get 8
            probe[name + "_ms"] = run_jobs(cs, 24 if len(cs) > 1 else 8, 910)
        if distributed:     # every rank must take the same path: the collectives are issued per context in turn
            for k in sorted(probe):
                t = dev.scalar(probe[k], torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                probe[k] = float(t.item())
        regime = min(REGIMES, key=lambda r: probe[r + "_ms"])
        # Tie rule.  three_jobs is the steady-state winner wherever the runtime gives its streams their own hardware queues
        # (34.4-35.4 ms against 35.7-36.4 with two) and the steadiest from run to run (the walk train moves between
        # 34.8 and 37.3 ms on one box, profiles/r04_world1_ab.log), but a short probe can put the regimes within noise of one
        # another on either side: another regime is taken only if it beats three_jobs by more than 1.5 %.
        if regime != "three_jobs" and "three_jobs" in REGIMES and probe["three_jobs_ms"] <= 1.015 * probe[regime + "_ms"]:
            regime = "three_jobs"
        probe["chosen"] = regime
    ctxs = apply_regime(regime)
    depth, lanes = REGIMES[regime]
    while len(pool) > depth:         # contexts the chosen regime does not use give their logs back
        pool.pop().close()
    info = ctxs[0].device_info()

    def barrier():
        for c in ctxs:
            c.sync()
        dev.sync()
        if distributed:
            dist.barrier()
        dev.sync()

    kernel_ms, steps_per_launch, stages = [], [], []

    def start(c, seed):
        c.zero_tally()
        c.launch(per_gpu, seed=seed, photon_offset=offset, f32_walk=args.f32_walk)

    def finish(c, record):
        """Complete the job in flight on c: reduce over ranks (N > 1; enqueued on the ctx stream), wait, read the
        96-byte counters."""
        if distributed:
            reduce_ctx(c)
        c.sync()
        st = note(c)
        if record:
            kernel_ms.append(c.last_kernel_ms())
            if st is not None:
                stages.append(st)
            steps_per_launch.append(c.read_counters()["steps"])   # rank 0: the reduced sum; part of "tally readback"

    def run_steps(n, seed0, record):
        for k in range(n):
            c = ctxs[k % depth]
            if k >= depth:
                finish(c, record)
            start(c, seed0 + k)
        for k in range(max(0, n - depth), n):     # drain in launch order
            finish(ctxs[k % depth], record)

    run_steps(args.warmup, 1000, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, 0, True)
    barrier()
    elapsed = time.perf_counter() - t0

    # How much of each job overlapped with its neighbours ON THIS RANK: span of a job's own HIP events (first kernel to
    # last) / wall time per job.  ~1: the jobs ran one after the other -- with depth >= 2 that means this rank's streams
    # shared a hardware queue (or the device was otherwise serialised); ~depth: they really were in flight together.
    own_overlap = float(np.mean(kernel_ms)) / (elapsed / args.steps * 1e3) if kernel_ms else 0.0
    overlap_ranks = [own_overlap]
    if distributed:
        t = dev.scalar(elapsed, torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ov = [dev.scalar(0.0, torch.float64) for _ in range(world)]
        dist.all_gather(ov, dev.scalar(own_overlap, torch.float64))
        overlap_ranks = [float(x.item()) for x in ov]
        km = dev.scalar(float(np.mean(kernel_ms)), torch.float64)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_avg_ms = float(km.item())
    else:
        kernel_avg_ms = float(np.mean(kernel_ms))

    # After the timed region, rank 0: one job ALONE on the device in the one_at_a_time regime (kernel durations nothing
    # overlaps: the per-kernel roofline figures) and the D2H readback of the grid into pinned host memory.
    alone, readback_ms = None, None
    if rank == 0 and not args.no_alone:
        c = ctxs[0]
        c.set_overlap(1); c.set_launch_config(0, 0)
        # the shipping default first (with the tail split where it applies: the walk's last photons finish in a second kernel
        # BESIDE the log reduction) = what one synchronous caller on one lane gets ...
        ms_default = []
        if hasattr(c, "set_tuning"):
            for k in range(3):
                c.zero_tally(); c.launch(per_gpu, seed=510 + k, photon_offset=offset, f32_walk=args.f32_walk); c.sync()
                ms_default.append(c.last_kernel_ms()); note(c)
            c.set_tuning("tail_split", 0)
        # ... then with the split off, so that every kernel really runs alone: these durations feed the per-kernel rooflines
        ms, st = [], None
        for k in range(3):
            c.zero_tally(); c.launch(per_gpu, seed=500 + k, photon_offset=offset, f32_walk=args.f32_walk); c.sync()
            ms.append(c.last_kernel_ms())
            st = note(c)
        if hasattr(c, "set_tuning"):
            c.set_tuning("tail_split", -1)
        alone = {"job_ms": float(np.mean(ms[1:]))}
        if ms_default:
            alone["job_ms_shipping_default"] = float(np.mean(ms_default[1:]))
        if st is not None:
            alone.update({k: st[k] for k in ("walk_ms", "scan_ms", "partition_ms", "reduce_ms")})
            alone["batches"] = st["batches"]
        nbytes = wl["grid"] ** 3 * REC_VALUE_BYTES[args.tally]
        buf = dev.host_buffer(nbytes)
        rb = []
        for k in range(3):
            c.sync(); t1 = time.perf_counter(); c.read_grid_into(buf); rb.append((time.perf_counter() - t1) * 1e3)
        readback_ms = float(min(rb[1:]))

    if rank == 0:
        total_steps = int(np.sum(steps_per_launch))          # reduced over ranks when distributed
        value = total_steps / elapsed
        steps_one_launch = total_steps / args.steps / world  # per rank per launch
        ms_per_step = elapsed / args.steps * 1e3
        # With work of several launches / lanes in flight a launch's own event-to-event time spans other kernels too;
        # the device completes one launch every ms_per_step, and that is the duration the algorithmic bytes are divided
        # by.  job_event_ms keeps the raw per-launch event time.
        launch_ms = kernel_avg_ms if regime == "one_at_a_time" else ms_per_step
        algo = steps_one_launch * BYTES_PER_STEP[args.tally]
        achieved = algo / (launch_ms * 1e-3) / 1e9
        sha = kernel_sources_sha()
        key = "%s_%s_%s_%s" % (args.workload, "f32walk" if args.f32_walk else "f64walk", args.tally, args.tally_mode)
        traffic, tsrc = None, {"file": "profiles/traffic.json", "key": key, "kernels_sha_now": sha}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(key)
            if ent:
                tsrc.update({k: ent.get(k) for k in ("tag", "head", "kernels_sha", "photons")})
                if ent.get("kernels_sha") == sha and ent.get("photons") == per_gpu:
                    traffic = ent["bytes"]
                else:
                    tsrc["stale"] = "PMC pass taken on other kernel sources / another size: not reported as this run's traffic"
        except Exception as e:      # no file: traffic stays null
            tsrc["error"] = str(e)[:80]
        out = {
            "metric": "photon_steps_per_sec", "value": value, "unit": "photon-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.f32_walk else "f64", "data": "synthetic",
            "photons_per_sec": world * per_gpu * args.steps / elapsed,
            "config": {"workload": wl["text"] % (per_gpu, wl["grid"], wl["voxel"]),
                       "tally": args.tally, "tally_mode": args.tally_mode, "rng": "rocRAND XORWOW, re-seeded per photon",
                       "regime": regime, "jobs_in_flight": depth, "lanes_per_launch": lanes, "regime_probe": probe,
                       "parallelism": "photon-id sharding x%d, RCCL reduce of the grid to rank 0 per step, enqueued on the "
                                      "job's stream" % world if world > 1 else "single GPU",
                       "reduce": {"backend": dist.get_backend() if distributed else None, "calls_rank0": n_reduces[0]},
                       "device": info["name"], "cus": info["cus"], "clock_mhz": info["clock_mhz"],
                       "hbm_gib": round(info["hbm_bytes"] / 2 ** 30, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                         "measured_gbs": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                         "kernel": "walk_kernel + k_log_scan/part/reduce (one job)" if args.tally_mode == "log"
                         else "walk_kernel", "kernel_ms": launch_ms, "job_event_ms": kernel_avg_ms,
                         "algorithmic_bytes_per_launch": algo},
        }
        if stages:
            rec = float(np.mean([x["records"] for x in stages]))
            rb_ = 4 + REC_VALUE_BYTES[args.tally]                              # bytes per deposit record in the log
            passes = 2 if wl["grid"] ** 3 > 1024 * 16384 else 1
            # algorithmic bytes of the partition: every record read once and written once in its final form.  (Grids of
            # more than 1024 tiles really move more: a 4-byte index read to count the digits, and a second pass over the
            # records of the tiles that are not hot -- that surplus counts against the kernel, not for it.)
            part_bytes = rec * (2 * rb_ - 2)
            out["roofline"]["deposit_records_per_launch"] = rec
            if hasattr(pool[0], "last_log_hot_tiles"):       # grids of > 1024 tiles: tiles that skip the second partition pass
                try:
                    out["roofline"]["hot_tiles"] = pool[0].last_log_hot_tiles()[0]
                except Exception:
                    pass
            out["roofline"]["kernels_overlapped_ms"] = {
                k: float(np.mean([x[k] for x in stages])) for k in ("walk_ms", "partition_ms", "reduce_ms")}
            out["roofline"]["kernels_overlapped_ms"]["note"] = (
                "sums of the kernels' own event-to-event times over the timed region; with several jobs / lanes in "
                "flight these overlap in time and are NOT per-step kernel times")
            if alone and "walk_ms" in alone:
                w, p_, r_ = alone["walk_ms"], alone["partition_ms"] + (alone.get("scan_ms", 0.0) if passes == 2 else 0.0), alone["reduce_ms"]
                out["roofline"]["kernels"] = [
                    {"kernel": "walk_kernel", "ms": w, "bound": "valu", "regime": "one job alone",
                     "note": "~350 VALU instr per photon-step (PMC SQ_INSTS_VALU); writes the %.1f GB deposit log" % (rec * rb_ / 1e9),
                     "photon_steps_per_sec": steps_one_launch / (w * 1e-3)},
                    {"kernel": "k_log_count1 + k_log_part<1> + k_log_count2 + k_log_part<2>" if passes == 2 else "k_log_part", "ms": p_, "bound": "hbm",
                     "regime": "one job alone", "achieved": part_bytes / (p_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": part_bytes / (p_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "algorithmic: every record read once (%d B) and written once (%d B: 2-byte in-tile position)%s" % (
                         rb_, rb_ - 2, "" if passes == 1 else
                         "; this grid has more than 1024 tiles: on top of that the digits are counted from the log (4 B index read) "
                         "and the records of the tiles that are not hot (lt_last_log_hot_tiles) take a second pass")},
                    {"kernel": "k_log_reduce", "ms": r_, "bound": "hbm", "regime": "one job alone",
                     "achieved": (rb_ - 2) * rec / (r_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (rb_ - 2) * rec / (r_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "algorithmic: every record read once (%d B)" % (rb_ - 2)}]
        if every["batches"]:
            passes_ = 2 if wl["grid"] ** 3 > 1024 * 16384 else 1
            out["roofline"]["all_dispatches_avg_ms"] = {
                "walk_kernel": every["walk_ms"] / every["batches"],
                "k_log_part (all passes of a batch)": every["partition_ms"] / every["batches"],
                "k_log_reduce": every["reduce_ms"] / every["batches"],
                "dispatches_per_kernel": every["batches"], "launches": every["launches"],
                "note": "mean over EVERY dispatch this process made (probe, warm-up, timed, reference launches; sub-batches "
                        "and pilot batches count as dispatches) of the HIP-event interval around the stage on its own stream.  "
                        "Compare with AverageNs of a rocprofv3 --kernel-trace --stats summary of the same command "
                        "(profiles/*_kernel_stats.csv) knowing what differs: an event interval also holds the time a kernel "
                        "waits for CUs that other streams' kernels occupy, so with jobs / lanes in flight it EXCEEDS the "
                        "kernel's own duration (measured, C2: walk 39.9 vs 38.0 ms, partition 13.7 vs 10.6, reduce 6.6 vs 4.2); "
                        "for a job alone the two agree (roofline.one_job_alone).  walk_kernel here = the rows "
                        "walk_kernel<.., 0> and <.., 1> there (whole walk / bulk of a split walk); the tail kernel <.., 2> "
                        "runs on its own stream and is in no stage sum; k_log_part<.., false> / <.., true> (beside a walk / "
                        "alone) are one stage here; %s" % (
                            "k_log_part<.,1,.> + k_log_count2 + k_log_part<.,2,.> are separate rows there (k_log_count1 belongs to the scan stage)" if passes_ == 2 else
                            "one partition dispatch per batch")}
        if alone:
            alone["note"] = ("one job alone on the device, one lane (default launch geometry, 4 waves/SIMD), 2 launches after the "
                             "timed region, tail split OFF: kernel durations nothing overlaps (the per-kernel rooflines); "
                             "job_ms_shipping_default = the same job as a caller gets it (tail split on where it applies)")
            out["roofline"]["one_job_alone"] = alone
        if readback_ms is not None:
            gb = wl["grid"] ** 3 * REC_VALUE_BYTES[args.tally] / 1e9
            out["readback"] = {"ms": readback_ms, "bytes": gb * 1e9, "gbs": gb / (readback_ms * 1e-3),
                               "note": "D2H of the raw grid into pinned host memory (one rank), outside the timed region",
                               "photon_steps_per_sec_incl_readback": steps_one_launch / ((launch_ms + readback_ms) * 1e-3),
                               "photons_per_sec_incl_readback": per_gpu / ((launch_ms + readback_ms) * 1e-3)}
        # ---- flat scalar copies of what the nested objects above hold (a consumer that keeps only scalar keys of
        # config / roofline still sees the probe, the kernels of one job alone and the stream overlap)
        cfg, rf = out["config"], out["roofline"]
        if probe:
            for k_, v_ in probe.items():
                cfg["probe_" + k_] = v_
        cfg["overlap_factor_min"], cfg["overlap_factor_max"] = min(overlap_ranks), max(overlap_ranks)
        cfg["overlap_factor_per_rank"] = ",".join("%.2f" % x for x in overlap_ranks)
        # a rank whose jobs in flight did not overlap (factor < 1.3 where depth >= 2 promises ~depth): its streams were serialised
        cfg["serialised_ranks"] = sum(1 for x in overlap_ranks if depth >= 2 and x < 1.3)
        cfg["gpu_max_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "")
        tag = args.workload
        rf[tag + "_steps_per_s"], rf[tag + "_ms"], rf[tag + "_frac"], rf[tag + "_regime"] = value / world, ms_per_step, achieved / HBM_PEAK_GBS, regime
        if alone:
            for k_ in ("job_ms", "job_ms_shipping_default", "walk_ms", "scan_ms", "partition_ms", "reduce_ms"):
                if k_ in alone:
                    rf["%s_alone_%s" % (tag, k_)] = alone[k_]
            if "job_ms_shipping_default" in alone:
                rf[tag + "_one_launch_ms"] = alone["job_ms_shipping_default"]
                rf[tag + "_one_launch_steps_per_s"] = steps_one_launch / (alone["job_ms_shipping_default"] * 1e-3)
        for kr in rf.get("kernels", []):
            if "frac" in kr:
                rf["%s_%s_frac" % (tag, "partition" if "part" in kr["kernel"] else "reduce")] = kr["frac"]
        if readback_ms is not None:
            rf[tag + "_readback_ms"] = readback_ms
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not fake:
            cpu = out["cpu_baseline"] = cpu_baseline(wl, args.workload)
            rf[tag + "_cpu_steps_per_s"] = cpu["value"]
        # ---- the other BASELINE configs (N = 1): their contexts come after this workload's have given their logs back
        extras = [] if (world != 1 or fake or distributed) else (
            [x for x in ("c3", "c4", "c5") if x != args.workload] if args.extras == "auto" and args.workload == "c2" and not args.photons
            else [x for x in args.extras.split(",") if x in WORKLOADS and x != args.workload] if args.extras not in ("auto", "none") else [])
        if extras:
            for c in pool:
                c.close()
            pool[:] = []
            for name in extras:
                try:
                    ex = measure_extra(name, make_ctx, args, 0 if args.no_cpu_baseline else 3.0)
                except Exception as e:      # an extra must never cost the headline its line
                    ex = {name + "_error": "%s: %s" % (type(e).__name__, str(e)[:160])}
                rf.update(ex)
                if cpu is not None and name + "_cpu_steps_per_s" in ex:
                    cpu[name + "_value"] = ex[name + "_cpu_steps_per_s"]
            rf["extras_note"] = ("cN_steps_per_s / cN_ms / cN_frac: whole jobs with two or three in flight (8 / 18 timed after 4 / 6 untimed, host clock "
                                 "between syncs; the fastest of the two_jobs, walk_train and (256^3 grids) three_jobs regimes, cN_regime), frac = 16 B x photon-steps / ms / 8 TB/s; cN_one_launch_*: one lt_launch alone with the "
                                 "library's defaults (mean of 3, device time); cN_alone_*: that launch's kernels on one lane with nothing "
                                 "beside them; cN_cpu_steps_per_s: the CPU oracle on cN_cpu_cores threads, a ~3 s sample")
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    for c in pool:
        c.close()


if __name__ == "__main__":
    main()
