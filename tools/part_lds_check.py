"""The LDS-DMA staged partition (lt_set_tuning "part_lds") against the register-staged one: the tally must be bit-identical;
per-stage device times of one launch, at the default grid and at the walk train's (3 workgroups per CU walk, the partition at
one workgroup per CU beside it).    python tools/part_lds_check.py [c2|c3|c4] [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 7
prob = {"c2": lambda: S.slab(n=256, voxel=0.1), "c3": lambda: S.two_layer(n=256, voxel=0.05)}[which]()
ctx = lt.Context(0)
for dtype in ("u64fx", "f64"):      # fixed point: bit-identical whatever the order; f64: the order of the adds inside a tile moves the last bits
  prob.apply(ctx, dtype); ctx.set_tally_mode("log"); ctx.set_overlap(1)
  ref = None
  print("tally", dtype)
  for label, knobs in (("registers", {}), ("lds-dma, 512 lanes", {"part_lds": 2}), ("registers, alone grid", {"part_alone": 1}),
                       ("lds-dma, 256 lanes (+ 256-lane reduce)", {"part_lds": 6})):
      with ctx.tuning(**knobs):
          best = None
          for r in range(3):
              ctx.zero_tally(); ctx.launch(n, seed=7); ctx.sync()
              st = ctx.last_log_stages()
              if r and (best is None or st["partition_ms"] < best["partition_ms"]): best = st
          t = np.asarray(ctx.read_grid_raw()).copy()
      if ref is None: ref = t
      same = bool((t == ref).all()) if dtype == "u64fx" else bool(np.allclose(t, ref, rtol=1e-12, atol=0))
      print("%-40s partition %6.2f ms  reduce %5.2f  walk %6.2f | tally identical to the first: %s  (sum %r)" % (
          label, best["partition_ms"], best["reduce_ms"], best["walk_ms"], same, float(t.sum())), flush=True)
      assert same, "the staged partition changed the tally"
ctx.close()
print("OK")
