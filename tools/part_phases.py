"""Clock sums per phase of the LDS-staged partition (k_log_part_lds), first wave of every workgroup -- needs the measurement
build:   bash tools/build_variant.sh pp -DLT_PART_PROF;  LT_HIP_LIBRARY=gpurun_ab/pp/liblt_hip.so python tools/part_phases.py [c2|c3]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
prob = {"c2": lambda: S.slab(n=256, voxel=0.1), "c3": lambda: S.two_layer(n=256, voxel=0.05)}[which]()
ctx = lt.Context(0)
prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1); ctx.set_tuning("part_lds", 2); ctx.set_tuning("tail_split", 0)
fn = lt.lib().lt_diag_part_phases
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 8)()
for r in range(3):
    ctx.zero_tally(); ctx.launch(10 ** 7, seed=r); ctx.sync()
    if r == 1: assert fn(out, 1) == 0      # reset after the warm-up launches
st, info = ctx.last_log_stages(), ctx.last_log_info()
assert fn(out, 0) == 0
names = ["top barrier", "stage issue + zero", "rank (LDS atomics)", "counts, cursor atomics, scan", "sorted copy", "cursor wait + bases", "write-out", "-"]
tot = sum(out[:7])
items = info["records"] / 4096
print("partition %.2f ms for %d records (%.0f items); clock sums of wave 0, %.0f clocks per item:" % (st["partition_ms"], info["records"], items, tot / items))
for n_, v in zip(names[:7], out[:7]):
    print("  %-30s %5.1f %%   %8.0f clocks per item" % (n_, 100.0 * v / tot, v / items))
ctx.close()
