"""Robustness run: BASELINE config 5's WHOLE job (1e8 photons, two-layer, 512^3) on one GPU -- log tally with one lane
and with two lanes per launch (many batches) against the atomic tally, fixed-point grids compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.two_layer(n=512, voxel=0.025)
n = 10 ** 8
res = {}
for mode, lanes in (("log", 1), ("log", 2), ("atomic", 1)):
    prob.apply(ctx, "u64fx"); ctx.set_tally_mode(mode); ctx.set_overlap(lanes)
    for rep in range(2):          # the first launch of a scene also measures its record rate (pilot) and allocates the logs
        ctx.zero_tally(); t0 = time.time(); ctx.launch(n, seed=11); ctx.sync(); dt = time.time() - t0
    c = ctx.read_counters(); info = ctx.last_log_info()
    res[(mode, lanes)] = ctx.read_grid_raw()
    print("%-6s lanes %d: wall %.3f s, device %.1f ms, %.2f Gsteps/s, steps %d, %s" % (
        mode, lanes, dt, ctx.last_kernel_ms(), c["steps"] / ctx.last_kernel_ms() / 1e6, c["steps"], info), flush=True)
    tot = (c["w_absorbed"] + c["w_lost_outside_grid"] + c["w_escaped_top"] + c["w_escaped_bottom"] + c["w_specular"]
           + c["w_roulette_net"] + c["w_capped"])
    print("   conservation residual / N = %.3e" % ((tot - n) / n), flush=True)
print("log (1 lane) == atomic (bit for bit):", np.array_equal(res[("log", 1)], res[("atomic", 1)]))
print("log (2 lanes) == atomic (bit for bit):", np.array_equal(res[("log", 2)], res[("atomic", 1)]))
