"""GPU sweep: launch geometry x precision x tally on C2, plus C3/C4/C5-per-GPU timings."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S

ctx = lt.Context(0)
def run(prob, n, dtype, f32, bpc=0, thr=0, reps=3, label="", lanes=1):
    prob.apply(ctx, dtype)
    ctx.set_launch_config(bpc, thr)
    ctx.set_overlap(lanes)
    best = 1e9
    for r in range(reps):
        ctx.zero_tally(); ctx.launch(n, seed=r, f32_walk=f32); ctx.sync()
        best = min(best, ctx.last_kernel_ms())
    c = ctx.read_counters()
    print("%-34s n=%.3g tally=%-5s walk=%s bpc=%d thr=%3d  %8.2f ms  %6.2f Gsteps/s  %5.1f steps/photon" % (
        label, n, dtype, "f32" if f32 else "f64", bpc, thr, best, c["steps"] / best / 1e6, c["steps"] / n), flush=True)

c2 = S.slab(n=256, voxel=0.1)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "geom"):
    for f32, dtype in ((False, "f64"), (True, "f32")):
        for thr in (64, 128, 256):
            for bpc in (1, 2, 3, 4, 6, 8):
                try:
                    run(c2, 10**7, dtype, f32, bpc, thr, label="C2 geometry")
                except Exception as e:
                    print("skip", bpc, thr, str(e)[:80])
if which in ("all", "tally"):
    for f32 in (False, True):
        for dtype in ("f32", "f64", "u64fx"):
            run(c2, 10**7, dtype, f32, label="C2 tally/precision")
if which in ("all", "configs"):
    run(S.slab(n=64, voxel=0.4), 10**4, "f64", False, label="C1 1e4 64^3")
    run(S.slab(n=64, voxel=0.4), 10**7, "f64", False, label="C1-geometry 1e7 64^3")
    run(S.two_layer(n=256, voxel=0.05), 10**7, "f64", False, label="C3 two-layer 256^3")
    run(S.two_layer(n=256, voxel=0.05), 10**7, "f32", True, label="C3 two-layer 256^3")
    run(S.cornell(256), 10**7, "f64", False, label="C4 cornell+cone mesh 256^3", reps=3)
    run(S.cornell(256), 10**7, "f32", True, label="C4 cornell+cone mesh 256^3", reps=3)
    run(S.two_layer(n=512, voxel=0.025), 12500000, "f64", False, label="C5 per-GPU share 512^3", reps=3)
    run(S.two_layer(n=512, voxel=0.025), 12500000, "f32", True, label="C5 per-GPU share 512^3", reps=3)
if which == "modes":
    for mode, lanes in (("atomic", 1), ("log", 1), ("log", 2)):
        ctx.set_tally_mode(mode)
        print("== tally mode", mode, "lanes per launch", lanes, flush=True)
        import functools
        run = functools.partial(run, lanes=lanes) if not isinstance(run, functools.partial) else functools.partial(run.func, lanes=lanes)
        run(S.slab(n=64, voxel=0.4), 10**7, "f64", False, label="C1-geometry 1e7 64^3")
        run(c2, 10**7, "f64", False, label="C2")
        run(c2, 10**7, "f32", True, label="C2")
        run(S.two_layer(n=256, voxel=0.05), 10**7, "f64", False, label="C3 two-layer 256^3")
        run(S.cornell(256), 10**7, "f64", False, label="C4 cornell+cone mesh 256^3")
        run(S.cornell(256), 10**7, "f32", True, label="C4 cornell+cone mesh 256^3")
        run(S.two_layer(n=512, voxel=0.025), 12500000, "f64", False, label="C5 per-GPU share 512^3")
        run(S.two_layer(n=512, voxel=0.025), 12500000, "f32", True, label="C5 per-GPU share 512^3")
