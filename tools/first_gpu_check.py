import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import light_transport_amd as lt

ctx = lt.Context(0)
print(ctx.device_info())
# RNG restatement vs rocRAND on device
for pid in (0, 1, 12345678901):
    a, b = ctx.rng_raw(42, pid, 64), O.rng_raw(42, pid, 64)
    print("rng", pid, np.array_equal(a, b))

media = [(0.1, 10.0, 0.9, 1.0)]
def setup(shape, voxel, dtype):
    ctx.set_media(media)
    ctx.set_layers([0, np.inf], [0])
    ctx.set_grid(shape, (-shape[0]*voxel/2, -shape[1]*voxel/2, 0.0), (voxel,)*3, dtype)
    ctx.set_source(0, (0, 0, 0), (0, 0, 1))
sc = O.OracleScene(media=media, grid_shape=(64,64,64), origin=(-12.8,-12.8,0), voxel=(0.4,)*3,
                   layers=dict(z_bounds=[0, np.inf], medium_idx=[0]))
# XORWOW f64 parity at C1
setup((64,64,64), 0.4, "f64")
ctx.launch(10000, seed=0); ctx.sync()
g = ctx.read_grid(); c = ctx.read_counters()
go, _, co = sc.run(10000, seed=0)
print("gpu", c); print("cpu", co)
d = np.abs(g-go); print("max abs diff", d.max(), "max rel", (d/(np.abs(go)+1e-300))[go>0].max(), "nviol", int((d > 1e-12+1e-9*np.abs(go)).sum()))
print("resid gpu", O.conservation_residual(c))
# table mode
rs = np.random.RandomState(0); tab = rs.rand(2000, 600, 4)
sc.max_steps = 1000000
ctx.zero_tally(); ctx.launch(2000, rng_table=tab); ctx.sync(); g = ctx.read_grid(); c = ctx.read_counters()
go, _, co = sc.run(2000, rng_table=tab)
d = np.abs(g-go); print("table: max abs diff", d.max(), "nviol", int((d > 1e-12+1e-9*np.abs(go)).sum()), c["steps"], co["steps"], c["w_capped"], co["w_capped"])
# u64fx bit parity
setup((64,64,64), 0.4, "u64fx")
ctx.launch(10000, seed=3); ctx.sync(); gfx = ctx.read_grid_raw()
_, gofx, _ = sc.run(10000, seed=3, want_fx=True)
print("u64fx equal:", np.array_equal(gfx, gofx), int((gfx != gofx).sum()))
# timing
for dtype, f32 in (("f64", False), ("f32", True), ("f32", False)):
    setup((256,256,256), 0.1, dtype)
    for n in (100000, 2000000):
        ctx.zero_tally(); ctx.launch(n, seed=1, f32_walk=f32); ctx.sync()
        ms = ctx.last_kernel_ms(); c = ctx.read_counters()
        print(dtype, "f32walk" if f32 else "f64walk", n, "ms", ms, "Gsteps/s", c["steps"]/ms/1e6, "resid", O.conservation_residual(c)/n)
