"""CPU suite: bench.py's N > 1 control flow at world size 2 over gloo with recording stand-in contexts
(tests/fake_ctx.py) -- the probe's rank agreement, the launch / reduce / readback order per regime, the JSON line.
The first 8-GPU run of the real thing must not be the first time this host logic executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(tmp_path, extra, nproc=2):
    log = os.path.join(str(tmp_path), "calls")
    env = dict(os.environ, LT_FAKE_LOG=log, PYTHONPATH=ROOT)
    port = 29700 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--backend", "gloo",
           "--ctx-factory", "tests.fake_ctx:make", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0: %r" % lines
    calls = [[json.loads(l) for l in open("%s.%d.jsonl" % (log, rk))] for rk in range(nproc)]
    return json.loads(lines[0]), calls


@pytest.mark.parametrize("regime,flags", [("two_jobs", ["--inflight", "2"]), ("three_jobs", ["--inflight", "3"]), ("walk_train", ["--inflight", "4"]),
                                          ("one_call", ["--inflight", "1"]),
                                          ("one_at_a_time", ["--inflight", "1", "--overlap", "1"]), ("probe", [])])
def test_bench_control_flow_world_size_2(tmp_path, regime, flags):
    K, W, n = 5, 2, 1000
    out, calls = run_bench(tmp_path, flags + ["--steps", str(K), "--warmup", str(W), "--photons", str(n)])
    assert out["metric"] == "photon_steps_per_sec" and out["n_gpus"] == 2 and out["steps"] == K and out["warmup"] == W
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "measured_gbs"}
    chosen = out["config"]["regime"]
    if regime == "probe":
        assert out["config"]["regime_probe"]["chosen"] == chosen
        assert {"one_call_ms", "two_jobs_ms", "three_jobs_ms", "walk_train_ms", "one_at_a_time_ms"} <= set(out["config"]["regime_probe"])
    else:
        assert chosen == regime and out["config"]["regime_probe"] is None
    # value = photon-steps of ALL ranks (reduced to rank 0) / wall: the fake job reports 281 * n + seed steps per rank
    timed_seeds = range(K)
    expect_steps = sum(2 * (281 * n + s) for s in timed_seeds)
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * K - expect_steps) < 1e-6 * expect_steps
    for rk, cl in enumerate(calls):
        # both ranks issue the same sequence of operations per context (collectives in the same order)
        def sig(x):      # (rank 0 alone runs six more jobs after the timed region: seeds 510.. shipping default, 500.. tail split off)
            return [(c["ctx"], c["op"]) for c in x if c["op"] == "reduce" or (c["op"] == "launch" and not 500 <= c["seed"] < 513)]
        assert sig(cl) == sig(calls[0])
        launches = [c for c in cl if c["op"] == "launch" and not 500 <= c["seed"] < 513]
        assert all(c["offset"] == rk * n and c["n"] == n for c in launches)          # disjoint photon-id ranges
        timed = [c for c in launches if c["seed"] < K]
        assert sorted(c["seed"] for c in timed) == list(timed_seeds)                   # EXACTLY K timed steps
        assert len([c for c in launches if 1000 <= c["seed"] < 1000 + W]) == W       # W warm-up steps
        reduces = [c for c in cl if c["op"] == "reduce"]
        probe_jobs = len([c for c in launches if 900 <= c["seed"] < 1000])             # the probe's jobs are reduced too
        assert len(reduces) == K + W + probe_jobs                                      # exactly one reduce per job
        # on every context: zero_tally -> launch -> reduce -> (sync) before the next launch
        state = {}
        for c in cl:
            if c["op"] == "launch":
                assert state.get(c["ctx"]) in (None, "idle", "zeroed"), (rk, c)
                state[c["ctx"]] = "flying"
            elif c["op"] == "reduce":
                assert state.get(c["ctx"]) == "flying"
            elif c["op"] == "sync":
                state[c["ctx"]] = "idle"
    depth = {"three_jobs": 3, "walk_train": 3, "two_jobs": 2, "one_call": 1, "one_at_a_time": 1}[chosen]
    assert out["config"]["jobs_in_flight"] == depth
    timed_ctx = {c["ctx"] for c in calls[0] if c["op"] == "launch" and c["seed"] < K}
    assert len(timed_ctx) == depth
    # rank 0's reference jobs: the split is switched off for the unoverlapped kernel times and restored afterwards
    tun = [(c["key"], c["value"]) for c in calls[0] if c["op"] == "set_tuning" and c["key"] == "tail_split"]
    assert tun == [("tail_split", 0), ("tail_split", -1)] and not any(c["op"] == "set_tuning" and c["key"] == "tail_split" for c in calls[1])
    # the walk train: its contexts -- and only they, in the timed region -- serialise their walks and launch at 3 workgroups per CU
    for cl in calls:
        last = {}
        for c in cl:
            if c["op"] == "set_tuning" and c["key"] == "serial_walks":
                last[c["ctx"]] = c["value"]
            if c["op"] == "set_launch_config":
                last[("bpc", c["ctx"])] = c["bpc"]
            if c["op"] == "launch" and c["seed"] < K:
                assert (last.get(c["ctx"], -1) == 1) == (chosen == "walk_train"), (chosen, c)
                assert last.get(("bpc", c["ctx"])) == {"walk_train": 3, "three_jobs": 2, "two_jobs": 2}.get(chosen, 0), (chosen, c)
    # only rank 0 reads the grid back, after the timed region
    assert any(c["op"] == "read_grid_into" for c in calls[0]) and not any(c["op"] == "read_grid_into" for c in calls[1])


def test_bench_self_launches_for_n_greater_than_1(tmp_path):
    """`python3 bench.py --gpus 2` from a plain shell (no RANK in the environment -- the shape of the driver's N = 1
    command): bench.py itself starts torch.distributed.run as a child and relays rank 0's JSON line and exit code."""
    log = os.path.join(str(tmp_path), "calls")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LT_FAKE_LOG=log, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--ctx-factory", "tests.fake_ctx:make",
           "--no-cpu-baseline", "--inflight", "2", "--steps", "3", "--warmup", "1", "--photons", "1000"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["reduce"] == {"backend": "gloo", "calls_rank0": 4}
    assert os.path.exists(log + ".0.jsonl") and os.path.exists(log + ".1.jsonl")      # two ranks really ran
    # a failing child fails the launcher too
    bad = subprocess.run([a if a != "tests.fake_ctx:make" else "tests.fake_ctx:no_such_factory" for a in cmd], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=240)
    assert bad.returncode != 0 and "no_such_factory" in bad.stderr
