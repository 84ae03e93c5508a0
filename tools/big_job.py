"""Robustness run: BASELINE config 5's WHOLE job (1e8 photons, two-layer, 512^3) on one GPU, log tally (many batches)
against atomic tally, fixed-point grids compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.two_layer(n=512, voxel=0.025)
n = 10 ** 8
res = {}
for mode in ("log", "atomic"):
    prob.apply(ctx, "u64fx"); ctx.set_tally_mode(mode)
    t0 = time.time(); ctx.launch(n, seed=11); ctx.sync(); dt = time.time() - t0
    c = ctx.read_counters(); st = ctx.last_log_stages()
    res[mode] = ctx.read_grid_raw()
    print(mode, "wall %.3f s, device %.1f ms, %.2f Gsteps/s, steps %d, batches %s" % (
        dt, ctx.last_kernel_ms(), c["steps"] / ctx.last_kernel_ms() / 1e6, c["steps"], st["batches"] if st else "-"), flush=True)
    tot = (c["w_absorbed"] + c["w_lost_outside_grid"] + c["w_escaped_top"] + c["w_escaped_bottom"] + c["w_specular"]
           + c["w_roulette_net"] + c["w_capped"])
    print("   conservation residual / N = %.3e" % ((tot - n) / n), flush=True)
print("log == atomic (bit for bit):", np.array_equal(res["log"], res["atomic"]))

# the same job through two contexts taking its batches in turn (JobPipeline.trace)
del ctx
def configure(c):
    prob.apply(c, "u64fx"); c.set_tally_mode("log")
pipe = lt.JobPipeline(configure, depth=2, raw=True)
def timed(batch):
    for c in pipe.ctxs: c.zero_tally(); c.sync()
    t0 = time.time(); done = k = 0
    while done < n:
        b = min(batch, n - done); pipe.ctxs[k % 2].launch(b, seed=11, photon_offset=done); done += b; k += 1
    for c in pipe.ctxs: c.sync()
    return time.time() - t0
timed(10 ** 7)                                               # warm-up: sizes the logs
for batch in (10 ** 7, 5 * 10 ** 6, 2500000):
    dt = timed(batch)
    steps = sum(c.read_counters()["steps"] for c in pipe.ctxs)
    g = pipe.ctxs[0].read_grid_raw() + pipe.ctxs[1].read_grid_raw()
    print("two contexts, batches of %.1e: %.3f s to the last sync, %.2f Gsteps/s, steps %d, == single launch: %s" % (
        batch, dt, steps / dt / 1e9, steps, np.array_equal(g, res["log"])), flush=True)
g, c = pipe.trace(n, seed=11, batch=5 * 10 ** 6)
print("JobPipeline.trace: steps %d, == single launch: %s" % (c["steps"], np.array_equal(g, res["log"])))
pipe.close()
