"""walk-alone time of the C2 slab (one lane, log tally), interleaved over builds given as LT libraries"""
import os, sys, subprocess
libs = sys.argv[1:]
code = r'''
import sys; sys.path.insert(0, "/root/repo")
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0); p = S.slab(n=256, voxel=0.1); p.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1)
w = []
for r in range(6):
    ctx.zero_tally(); ctx.launch(10**7, seed=r); ctx.sync()
    if r: w.append(ctx.last_log_stages()["walk_ms"])
print("walk ms: min %.2f median %.2f" % (min(w), sorted(w)[len(w)//2]))
'''
for rep in range(2):
    for l in libs:
        env = dict(os.environ); env["LT_HIP_LIBRARY"] = "" if l == "tree" else l
        if l == "tree": env.pop("LT_HIP_LIBRARY")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("%-28s %s" % (l, (out.stdout.strip() or out.stderr.strip()[-200:])), flush=True)
