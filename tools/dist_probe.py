"""torchrun world-1 probe: where does the per-step time go when a collective sits between two jobs in flight?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import light_transport_amd as lt
from light_transport_amd import distributed as ltd
from tests import scenes as S

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
prob = S.slab(n=256, voxel=0.1)
n, steps, depth = 10 ** 7, 10, 2
ctxs = [lt.Context(0) for _ in range(depth)]
for c in ctxs:
    prob.apply(c, "f64"); c.set_tally_mode(1); c.set_launch_config(2, 256); c.reserve_log(n)
    c.zero_tally(); c.launch(n, seed=99); c.sync()
t = ltd.device_grid_tensor(ctxs[0]); dist.reduce(t, dst=0); torch.cuda.synchronize()

def variant(name, fin):
    acc = dict(sync=0.0, coll=0.0, wait=0.0)
    t0 = time.perf_counter()
    for k in range(steps):
        c = ctxs[k % depth]
        if k >= depth: fin(c, acc)
        c.zero_tally(); c.launch(n, seed=k)
    for k in range(steps - depth, steps): fin(ctxs[k % depth], acc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.2f ms/step   host ms/step: sync %.2f  collective calls %.2f  stream wait %.2f" % (
        name, dt / steps * 1e3, acc["sync"] / steps * 1e3, acc["coll"] / steps * 1e3, acc["wait"] / steps * 1e3), flush=True)

def f_sync(c, a):
    t = time.perf_counter(); c.sync(); a["sync"] += time.perf_counter() - t
def f_reduce(c, a):
    t = time.perf_counter(); c.sync(); a["sync"] += time.perf_counter() - t
    t = time.perf_counter()
    for x in (ltd.device_grid_tensor(c),) + ltd.device_counter_tensors(c): dist.reduce(x, dst=0)
    a["coll"] += time.perf_counter() - t
    t = time.perf_counter(); torch.cuda.current_stream(0).synchronize(); a["wait"] += time.perf_counter() - t
def f_reduce_grid_only(c, a):
    t = time.perf_counter(); c.sync(); a["sync"] += time.perf_counter() - t
    t = time.perf_counter(); dist.reduce(ltd.device_grid_tensor(c), dst=0); a["coll"] += time.perf_counter() - t
    t = time.perf_counter(); torch.cuda.current_stream(0).synchronize(); a["wait"] += time.perf_counter() - t
def f_tensor_only(c, a):
    t = time.perf_counter(); c.sync(); a["sync"] += time.perf_counter() - t
    t = time.perf_counter(); x = ltd.device_grid_tensor(c); y = ltd.device_counter_tensors(c); a["coll"] += time.perf_counter() - t

variant("ctx.sync only", f_sync)
variant("tensor views only", f_tensor_only)
variant("reduce grid only", f_reduce_grid_only)
variant("reduce grid + counters", f_reduce)
dist.barrier(); dist.destroy_process_group()
