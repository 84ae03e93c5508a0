#!/usr/bin/env python3
"""Headline benchmark of the photon-transport hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete job of config C2 on every rank: zero the tally, trace
1e7 photons (homogeneous semi-infinite slab mu_a=0.1 mu_s=10 g=0.9 n=1, 256^3 grid
of 0.1 mm voxels, pencil beam, f64 walk, XORWOW, f64 tally; deposits go through
the log-structured tally: walk -> deposit log -> tile partition -> LDS reduce) and
-- for N > 1 -- sum-reduce the voxel grid + counters to rank 0 with RCCL.  Ranks trace disjoint
photon-id ranges (weak scaling: 1e7 photons per GPU); there is no other
collective.  Inputs are synthetic by nature (the scene is ~100 bytes of constants,
resident in HBM before the timed region).  By default two jobs are in flight per
GPU (after a short untimed probe has confirmed that they overlap; --inflight 2
forces it, --inflight 1 forbids it: two contexts, i.e. two HIP streams with their own grid and
deposit log, take the steps in turn), so that the bandwidth-bound log reduction of
one job runs beside the VALU-bound walk of the next; every step is still a
complete job and all K of them finish inside the timed region.

Prints ONE JSON line on rank 0:  metric = photon-steps/s over all ranks, plus
  roofline      algorithmic tally bytes (16 B per photon-step for the f64 tally:
                8 B read + 8 B write of one voxel) per launch / the time the device
                takes per launch (wall / K with jobs in flight; the job's event time
                with --inflight 1), against the 8 TB/s HBM peak; per-kernel
                durations from HIP events on each ctx's own stream;
  cpu_baseline  the CPU oracle (oracle/, a port -- the reference has no such
                path) on all host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Jobs in flight need their streams on DIFFERENT hardware queues: the HIP runtime multiplexes a process's streams
# onto GPU_MAX_HW_QUEUES (default 4) queues, and once torch + RCCL have taken theirs two contexts can end up sharing
# one, which serialises their kernels (measured: 57 instead of 39 ms per step under torch.distributed.run).  Must be
# set before the runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

N_PHOTONS = 10 ** 7
GRID_N, VOXEL = 256, 0.1
MEDIUM = (0.1, 10.0, 0.9, 1.0)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_STEP = {"f32": 8, "f64": 16, "u64fx": 16}


def configure(ctx, tally):
    half = GRID_N * VOXEL / 2
    ctx.set_media([MEDIUM])
    ctx.set_layers([0.0, np.inf], [0], 1.0, 1.0)
    ctx.set_grid((GRID_N,) * 3, (-half, -half, 0.0), (VOXEL,) * 3, tally)
    ctx.set_source(0, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0))


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


def cpu_baseline(target_seconds=12.0):
    """CPU oracle (port) on every host core, bounded sample of C2."""
    from oracle import oracle as O
    cores = effective_cores()
    half = GRID_N * VOXEL / 2
    sc = O.OracleScene([MEDIUM], (GRID_N,) * 3, (-half, -half, 0.0), (VOXEL,) * 3,
                       layers=dict(z_bounds=[0.0, np.inf], medium_idx=[0]))
    t0 = time.perf_counter()
    _, _, c = sc.run(50000, seed=0, threads=cores)
    probe = time.perf_counter() - t0
    rate = c["steps"] / probe
    n = int(min(N_PHOTONS, max(100000, target_seconds * rate / 281.0)))
    t0 = time.perf_counter()
    _, _, c = sc.run(n, seed=0, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": c["steps"] / dt, "unit": "photon-steps/s", "cores": cores, "kind": "port",
            "sample": "first %d photons of the same C2 workload (f64, 256^3 f64 grid), %.1f s wall, pthreads" % (n, dt),
            "photons_per_sec": n / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--photons", type=int, default=N_PHOTONS, help="photons per GPU per step (default: C2's 1e7)")
    ap.add_argument("--tally", default="f64", choices=["f32", "f64", "u64fx"])
    ap.add_argument("--f32-walk", action="store_true", help="f32 walk arithmetic (default f64, the reference's dtype)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--tally-mode", default="log", choices=["log", "atomic", "auto"],
                    help="log: deposit log + tile partition + LDS reduce (default); atomic: one global atomic per deposit")
    ap.add_argument("--inflight", type=int, default=0,
                    help="jobs in flight per GPU (contexts taking the steps in turn); 1 = strictly one job at a time; "
                         "0 (default) = 2 if a short untimed probe confirms that two jobs overlap on this device, else 1")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the single-job reference launches after the timed region (keeps a rocprofv3 kernel "
                         "trace of this command to launches of the timed regime only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "RANK" in os.environ
    if args.gpus != world and distributed:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and not distributed:
        raise SystemExit("for N > 1 launch through torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import light_transport_amd as lt
    from light_transport_amd import distributed as ltd
    # --inflight D: D contexts (stream + grid + deposit log each) take the steps in turn, so the bandwidth-bound log
    # reduction of one job runs beside the VALU-bound walk of the next.  The walk is then launched at 2 workgroups per
    # CU per job: two walks together fill the 4 waves/SIMD the register file holds, one walk leaves room for the
    # other job's partition / reduce workgroups.
    auto_depth = args.inflight <= 0
    depth = 2 if auto_depth else args.inflight

    def walk_bpc(d):
        return args.blocks_per_cu or ((3 if args.f32_walk else 2) if d > 1 else 0)   # f32 walk: 96 VGPRs, 5 waves/SIMD fit

    def set_geometry(c, d):
        b = walk_bpc(d)
        c.set_launch_config(b, args.threads or (256 if b else 0))

    ctxs = []
    for _ in range(depth):
        c = lt.Context(local_rank)
        configure(c, args.tally)
        c.set_tally_mode(args.tally_mode)
        set_geometry(c, depth)
        if args.tally_mode != "atomic":
            c.reserve_log(args.photons)   # scratch allocation is set-up, not part of a step (matters when --warmup 0)
        ctxs.append(c)
    probe = None
    if auto_depth:
        # Untimed probe: does the device really run two jobs side by side?  (It does not when the two contexts' streams
        # share a hardware queue, see GPU_MAX_HW_QUEUES above.)  One job at a time on ctx 0 against four jobs in turn.
        def run_jobs(cs, n):
            for c in cs:
                c.sync()
            t0 = time.perf_counter()
            for k in range(n):
                c = cs[k % len(cs)]
                if k >= len(cs):
                    c.sync()
                c.zero_tally(); c.launch(args.photons, seed=900 + k, photon_offset=rank * args.photons, f32_walk=args.f32_walk)
            for c in cs:
                c.sync()
            return (time.perf_counter() - t0) / n * 1e3
        set_geometry(ctxs[0], 1)
        run_jobs(ctxs[:1], 1)
        t_one = run_jobs(ctxs[:1], 2)
        set_geometry(ctxs[0], 2)
        run_jobs(ctxs, 2)
        t_two = run_jobs(ctxs, 4)
        use_two = t_two < 0.97 * t_one
        if distributed:     # every rank must take the same path: the collectives are issued per context in turn
            flag = torch.tensor([1 if use_two else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_two = bool(flag.item())
        probe = {"one_at_a_time_ms": t_one, "two_in_flight_ms": t_two, "chosen": 2 if use_two else 1}
        if not use_two:
            ctxs.pop().close()
            depth = 1
            set_geometry(ctxs[0], 1)
    bpc = walk_bpc(depth)
    info = ctxs[0].device_info()
    per_gpu = args.photons
    offset = rank * per_gpu     # disjoint id ranges; streams depend on (seed, id) only

    def barrier():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms, steps_per_launch, stages = [], [], []

    def start(c, seed):
        c.zero_tally()
        c.launch(per_gpu, seed=seed, photon_offset=offset, f32_walk=args.f32_walk)

    def finish(c, record):
        """Complete the job in flight on c: wait, reduce over ranks (N > 1), read the 96-byte counters."""
        if distributed:
            ltd.reduce_device(c, dst=0)       # RCCL sum of grid + counters to rank 0
        else:
            c.sync()
        if record:
            kernel_ms.append(c.last_kernel_ms())
            st = c.last_log_stages()
            if st is not None and st["batches"] == 1:
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

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device="cuda")
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_avg_ms = float(km.item())
    else:
        kernel_avg_ms = float(np.mean(kernel_ms))

    # one job alone on the device, default launch geometry, outside the timed region: the latency of a single job
    # and kernel durations that no other job overlaps (context for the per-kernel figures above)
    alone = None
    if rank == 0 and depth > 1 and not args.no_alone:
        c = ctxs[0]
        set_geometry(c, 1)
        ms = []
        for k in range(3):
            c.zero_tally(); c.launch(per_gpu, seed=500 + k, photon_offset=offset, f32_walk=args.f32_walk); c.sync()
            ms.append(c.last_kernel_ms())
        st = c.last_log_stages()
        alone = {"job_ms": float(np.mean(ms[1:]))}
        if st is not None:
            alone.update({k: st[k] for k in ("walk_ms", "partition_ms", "reduce_ms")})

    if rank == 0:
        total_steps = int(np.sum(steps_per_launch))          # reduced over ranks when distributed
        value = total_steps / elapsed
        steps_one_launch = total_steps / args.steps / world  # per rank per launch
        ms_per_step = elapsed / args.steps * 1e3
        # With D jobs in flight a launch's own event-to-event time spans the other job's kernels too; the device
        # completes one launch every ms_per_step, and that is the duration the algorithmic bytes are divided by.
        # job_event_ms keeps the raw per-launch event time (what rocprofv3's kernel trace adds up to).
        launch_ms = ms_per_step if depth > 1 else kernel_avg_ms
        achieved = steps_one_launch * BYTES_PER_STEP[args.tally] / (launch_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("%s_%s_%s" % ("f32walk" if args.f32_walk else "f64walk", args.tally,
                                                              args.tally_mode))
            except Exception:
                traffic = None
        out = {
            "metric": "photon_steps_per_sec", "value": value, "unit": "photon-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.f32_walk else "f64", "data": "synthetic",
            "photons_per_sec": world * per_gpu * args.steps / elapsed,
            "config": {"workload": "C2: %.0e photons per GPU, homogeneous semi-infinite slab (mu_a=0.1, mu_s=10, g=0.9, "
                                   "n=1), %d^3 voxel grid (%.1f mm), pencil beam" % (per_gpu, GRID_N, VOXEL),
                       "tally": args.tally, "tally_mode": args.tally_mode, "rng": "rocRAND XORWOW, re-seeded per photon",
                       "jobs_in_flight": depth, "walk_workgroups_per_cu": bpc or "occupancy", "inflight_probe": probe,
                       "parallelism": "photon-id sharding x%d, RCCL reduce of the grid to rank 0 per step" % world
                       if world > 1 else "single GPU", "device": info["name"], "cus": info["cus"],
                       "clock_mhz": info["clock_mhz"], "hbm_gib": round(info["hbm_bytes"] / 2 ** 30, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "walk_kernel + k_log_scan/part/reduce (one job)" if args.tally_mode == "log"
                         else "walk_kernel", "kernel_ms": launch_ms, "job_event_ms": kernel_avg_ms,
                         "algorithmic_bytes_per_launch": steps_one_launch * BYTES_PER_STEP[args.tally]},
        }
        if stages:
            # per-kernel figures of the job, live from HIP events on each ctx's stream over the timed region
            rec = float(np.mean([x["records"] for x in stages]))
            rb = 4 + {"f32": 4, "f64": 8, "u64fx": 8}[args.tally]          # bytes per deposit record in the log
            pb = 2 * rb - 2                                                    # partition: read rb, write rb - 2
            w, p_, r_ = (float(np.mean([x[k] for x in stages])) for k in ("walk_ms", "partition_ms", "reduce_ms"))
            shared = " (shares the device with the other job in flight: durations overlap)" if depth > 1 else ""
            out["roofline"]["kernels"] = [
                {"kernel": "walk_kernel", "ms": w, "bound": "valu",
                 "note": "349 VALU instr per photon-step (PMC SQ_INSTS_VALU, profiles/r01e_pmc_sq.csv); writes the %.1f GB deposit log%s"
                         % (rec * rb / 1e9, shared),
                 "photon_steps_per_sec": steps_one_launch / (w * 1e-3)},
                {"kernel": "k_log_part", "ms": p_, "bound": "hbm", "achieved": pb * rec / (p_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": pb * rec / (p_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "note": "algorithmic: every record read once (%d B) and written once (%d B: 2-byte in-tile position)" % (rb, rb - 2) + shared},
                {"kernel": "k_log_reduce", "ms": r_, "bound": "hbm", "achieved": (rb - 2) * rec / (r_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": (rb - 2) * rec / (r_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "note": "algorithmic: every record read once (%d B)" % (rb - 2) + shared}]
            out["roofline"]["deposit_records_per_launch"] = rec
        if alone:
            alone["note"] = ("one job alone on the device (default launch geometry, 4 waves/SIMD), 2 launches after the timed "
                             "region: single-job latency and kernel durations nothing overlaps")
            if "partition_ms" in alone and stages:
                alone["k_log_part_frac"] = pb * rec / (alone["partition_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                alone["k_log_reduce_frac"] = (rb - 2) * rec / (alone["reduce_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["one_job_alone"] = alone
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
