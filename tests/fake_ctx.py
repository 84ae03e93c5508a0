"""Recording stand-in for light_transport_amd.Context: lets the CPU suite drive bench.py's real control flow at world
size 2 over gloo (regime probe, rank agreement, start / finish order, collective order, rank-0 readback) without a GPU.
Every call is appended to LT_FAKE_LOG.<rank>.jsonl; reduce_to issues a real gloo reduce so that a rank-dependent
order of collectives would dead-lock (and time the test out) instead of passing silently."""
import json
import os

import numpy as np

_n = [0]


class FakeContext:
    def __init__(self):
        self.id = _n[0]; _n[0] += 1
        self.rank = int(os.environ.get("RANK", "0"))
        self.path = "%s.%d.jsonl" % (os.environ["LT_FAKE_LOG"], self.rank)
        self.steps = 0          # counters of the job in flight
        self.launched = None
        self.lanes = 0
        self._log("create")

    def _log(self, what, **kw):
        with open(self.path, "a") as f:
            f.write(json.dumps(dict(ctx=self.id, op=what, **kw)) + "\n")

    # scene / configuration
    def set_media(self, m): self._log("set_media", n=len(m))
    def set_layers(self, z, idx, a, b): self._log("set_layers", n=len(idx))
    def set_grid(self, shape, origin, voxel, dtype): self.shape = tuple(shape); self._log("set_grid", shape=list(shape), dtype=dtype)
    def set_source(self, *a): self._log("set_source")
    def set_tally_mode(self, mode, log_bytes=0): self._log("set_tally_mode", mode=mode)
    def set_overlap(self, lanes): self.lanes = lanes; self._log("set_overlap", lanes=lanes)
    def set_launch_config(self, b, t): self._log("set_launch_config", bpc=b, threads=t)
    def reserve_log(self, n): self._log("reserve_log", n=n)
    def set_tuning(self, key, value=-1): self._log("set_tuning", key=key, value=value)

    # run
    def zero_tally(self):
        assert self.launched is None, "zero_tally while a job is in flight on this context"
        self.steps = 0; self._log("zero_tally")

    def launch(self, n, seed=0, photon_offset=0, f32_walk=False):
        assert self.launched is None, "launch while a job is in flight on this context"
        self.launched = dict(n=n, seed=seed, offset=photon_offset)
        self.steps = 281 * n + seed          # a deterministic "photon-step count" of this job
        self._log("launch", n=n, seed=seed, offset=photon_offset)

    def sync(self):
        self.launched = None; self._log("sync")

    def reduce_to(self, dist, dst):
        import torch
        assert self.launched is not None, "reduce without a job in flight"
        t = torch.tensor([self.steps], dtype=torch.int64)
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
        if self.rank == dst:
            self.steps = int(t.item())
        self._log("reduce", dst=dst)

    def last_kernel_ms(self): return 40.0
    def last_log_stages(self): return dict(walk_ms=30.0, scan_ms=0.1, partition_ms=10.0, reduce_ms=3.0, records=1800, batches=1)
    def read_counters(self): self._log("read_counters", steps=self.steps); return dict(steps=self.steps)
    def read_grid_into(self, buf): self._log("read_grid_into", nbytes=int(np.asarray(buf).nbytes)); return buf
    def device_info(self): return dict(name="fake", cus=256, clock_mhz=2400, hbm_bytes=288 * 2 ** 30)
    def close(self): self._log("close")


def make():
    return FakeContext()
