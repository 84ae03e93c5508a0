"""Two contexts on one GPU, steps alternating between them (each context has its own stream, log and grid): does the
log reduction of step k overlap the walk of step k+1?  Prints ms per step for depth 1 (serial) and depth 2.
    python tools/overlap_probe.py [steps] [photons] [DxB,...]      (D contexts, walk at B workgroups per CU; 0 = all)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 7
# LT_PROBE_SCENE=c4: the Cornell cavity + cone (mesh walk) instead of the C2 slab
prob = S.cornell(256) if os.environ.get("LT_PROBE_SCENE") == "c4" else S.slab(n=256, voxel=0.1)


def run(depth, tally="f64", f32=False, bpc=0):
    ctxs = [lt.Context(0) for _ in range(depth)]
    for c in ctxs:
        prob.apply(c, tally)
        c.set_tally_mode(1)
        c.set_overlap(1)
        c.set_launch_config(bpc, 256 if bpc else 0)
        c.reserve_log(n)
        c.zero_tally(); c.launch(n, seed=99, f32_walk=f32); c.sync()      # warm-up: sizes the log
    t0 = time.perf_counter()
    tot = 0
    for k in range(steps):
        c = ctxs[k % depth]
        if k >= depth:
            c.sync(); tot += c.read_counters()["steps"]
        c.zero_tally(); c.launch(n, seed=k, f32_walk=f32)
    for c in ctxs:
        c.sync(); tot += c.read_counters()["steps"]
    dt = time.perf_counter() - t0
    print("depth %d  walk blocks/CU %d  %s walk %s tally: %.2f ms per step, %.2f G photon-steps/s" % (
        depth, bpc, "f32" if f32 else "f64", tally, dt / steps * 1e3, tot / dt / 1e9), flush=True)
    for c in ctxs:
        c.close()


cfg = sys.argv[3] if len(sys.argv) > 3 else "3x3,2x2"        # depth x blocks/CU pairs
for pair in cfg.split(","):
    d, b = pair.split("x")
    run(int(d), bpc=int(b))
