"""Several photon jobs in flight on one GPU.

A ``Context`` is one HIP stream with its own voxel grid and deposit log, and ``launch`` is asynchronous.  The walk is
compute-bound, the log reduction that follows it is bandwidth-bound; driven one job at a time they run back to back.
``JobPipeline`` keeps ``depth`` identically configured contexts and hands consecutive jobs to them in turn, with the walk
limited to 2 workgroups per CU per job, so that one job's reduction runs beside the next job's walk (C2 on MI355X: 37.5 ms
per job instead of 45.5).  Results come back in submission order and are the same numbers a single
context produces: a photon's trajectory depends on (seed, photon id) only.

The reference's counterpart is Numba's thread pool working through the pixels of one render_scene call
(path_tracing_fix1.py:144-148); here the unit of concurrency is a whole job.

In a process that also runs torch / RCCL, export GPU_MAX_HW_QUEUES=8 before the HIP runtime starts: the runtime maps a
process's streams onto 4 hardware queues by default and two contexts that share one are serialised (bench.py does this).
"""
import collections

from . import _lib


class JobPipeline:
    def __init__(self, configure, device_id=0, depth=2, walk_workgroups_per_cu=2, raw=False):
        """``configure(ctx)`` sets media, geometry, grid and source on a fresh context; it is called ``depth`` times.
        raw=True returns the tally in its own dtype (f32 / f64 / u64 fixed point) instead of float64."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.ctxs = []
        for _ in range(depth):
            c = _lib.Context(device_id)
            configure(c)
            if depth > 1 and walk_workgroups_per_cu:
                c.set_launch_config(walk_workgroups_per_cu, 256)
            self.ctxs.append(c)
        self.raw = bool(raw)
        self._pending = collections.deque()       # (ctx, tag) in submission order
        self._next = 0

    def submit(self, n_photons, seed=0, photon_offset=0, f32_walk=False, tag=None):
        """Start a job (zeroed tally) on the next context.  If that context still holds an unfinished job it is
        completed first and returned as (tag, grid, counters); otherwise None."""
        c = self.ctxs[self._next % len(self.ctxs)]
        self._next += 1
        done = None
        if self._pending and len(self._pending) >= len(self.ctxs):
            done = self._collect()
        c.zero_tally()
        c.launch(n_photons, seed=seed, photon_offset=photon_offset, f32_walk=f32_walk)
        self._pending.append((c, tag))
        return done

    def _collect(self):
        c, tag = self._pending.popleft()
        c.sync()
        return tag, (c.read_grid_raw() if self.raw else c.read_grid()), c.read_counters()

    def drain(self):
        """Complete every job still in flight, in submission order."""
        out = []
        while self._pending:
            out.append(self._collect())
        return out

    def run(self, jobs):
        """jobs: iterable of dicts with the keyword arguments of ``submit``.  Yields (tag, grid, counters) per job,
        in order."""
        for j in jobs:
            done = self.submit(**j)
            if done is not None:
                yield done
        for done in self.drain():
            yield done

    def trace(self, n_photons, seed=0, photon_offset=0, batch=None, f32_walk=False):
        """ONE large job spread over the contexts: ids [photon_offset, photon_offset + n_photons) are cut into batches
        of ``batch`` photons (default: n / (4 * depth)) that the contexts take in turn, each ACCUMULATING into its own
        grid, so that one batch's log reduction runs beside the next batch's walk.  Returns (grid, counters) summed
        over the contexts -- identical to a single launch for the integer tally and the counters' integer fields,
        equal up to summation order otherwise.  (Since lt_set_overlap a single ``launch`` does this by itself, on two
        streams of ONE context and into one grid; ``trace`` remains for hosts that want the batches on separate
        contexts.)"""
        if self._pending:
            raise RuntimeError("trace() needs an idle pipeline: drain() first")
        n_photons = int(n_photons)
        if batch is None:
            batch = max(1, -(-n_photons // (4 * len(self.ctxs))))
        for c in self.ctxs:
            c.zero_tally()
        done, k = 0, 0
        while done < n_photons:
            b = min(int(batch), n_photons - done)
            self.ctxs[k % len(self.ctxs)].launch(b, seed=seed, photon_offset=photon_offset + done, f32_walk=f32_walk)
            done += b
            k += 1
        grid, counters = None, None
        for c in self.ctxs:
            c.sync()
            g = c.read_grid_raw() if self.raw else c.read_grid()
            cn = c.read_counters()
            if grid is None:
                grid, counters = g, dict(cn)
            else:
                grid += g
                for key, v in cn.items():
                    counters[key] += v
        return grid, counters

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []
