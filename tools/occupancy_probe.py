"""How does a lone walk scale with its resident workgroups?  C2, one lane, atomic-free log mode, walk_ms of one launch at 1 / 2 / 3 / 4
workgroups per CU (lt_set_launch_config).  Even spread over the CUs vs packing shows in the 2-per-CU time (even: ~1.5x the full
time, the SIMDs half full; packed onto half of the CUs: 2x).   python tools/occupancy_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.slab(n=256, voxel=0.1)
prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1); ctx.set_tuning("tail_split", 0)
for bpc in (4, 3, 2, 1, 2, 4):
    ctx.set_launch_config(bpc, 256)
    best = 1e9
    for r in range(3):
        ctx.zero_tally(); ctx.launch(10 ** 7, seed=r); ctx.sync()
        best = min(best, ctx.last_log_stages()["walk_ms"])
    print("walk at %d workgroups per CU: %.2f ms" % (bpc, best), flush=True)
