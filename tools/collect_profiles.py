"""Copy the rocprof summaries of the last tools/gpu_profile.sh run into profiles/<tag>_* and refresh
profiles/traffic.json.  Counters are taken from the LAST job of each PMC pass: the last walk_kernel dispatch and the
k_log_* dispatches that follow it (one complete job, one lane: walk + scan + partition pass(es) + reduce).
    python tools/collect_profiles.py <tag> [workload=c2] [photons]"""
import csv, glob, json, os, shutil, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
photons = int(float(sys.argv[3])) if len(sys.argv) > 3 else bench.WORKLOADS[workload]["photons"]
key = "%s_f64walk_f64_log" % workload
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
def newest(pat):
    f = sorted(glob.glob(os.path.join(G, pat), recursive=True), key=os.path.getmtime)   # gpurun_out keeps older runs
    return f[-1] if f else None
ks = newest("prof_trace_%s/**/*_kernel_stats.csv" % workload)
if ks:
    shutil.copy(ks, os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, workload)))
job = collections.defaultdict(float)                                    # counter -> sum over the job's kernels
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("fetch", "write", "sq"):
    f = newest("prof_%s_%s/**/*_counter_collection.csv" % (d, workload))
    if not f:
        continue
    shutil.copy(f, os.path.join(P, "%s_%s_pmc_%s.csv" % (tag, workload, d)))
    rows = [r for r in csv.DictReader(open(f)) if "walk_kernel" in r["Kernel_Name"] or "k_log_" in r["Kernel_Name"] or "k_grid_add" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # the PMC passes run `bench.py --inflight 1 --overlap 1 --steps 2 --warmup 1 --no-alone`: 3 launches of b batches
    # each (+ the pilot batch of the first): the last launch = the last b walk dispatches and everything after the
    # first of them
    walks = [int(r["Dispatch_Id"]) for r in rows if "walk_kernel" in r["Kernel_Name"] and r["Counter_Name"] == rows[0]["Counter_Name"]]
    b = max(1, (len(walks) - 1) // 3)
    last_walk = sorted(walks)[-b]
    for r in rows:
        if int(r["Dispatch_Id"]) >= last_walk:
            kn = r["Kernel_Name"]
            name = kn[kn.index("k_log_"):].split("(")[0] if "k_log_" in kn else kn.split("(")[0].split("::")[-1]
            job[r["Counter_Name"]] += float(r["Counter_Value"])
            per_kernel[name][r["Counter_Name"]] += float(r["Counter_Value"])
bl = os.path.join(G, "bench_%s.log" % workload)
if os.path.exists(bl):
    line = [l for l in open(bl) if l.startswith("{")][-1]
    open(os.path.join(P, "%s_bench_%s_n1.json" % (tag, workload)), "w").write(line)
tf = os.path.join(P, "traffic.json")
t = json.load(open(tf)) if os.path.exists(tf) else {}
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
# MI355X_MICROARCH "HBM": on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads (doubled here);
# WRITE_SIZE is exact for 16-B-per-lane streaming stores and float atomics; both in KiB.
t[key] = {
    "bytes": (2 * job.get("FETCH_SIZE", 0.0) + job.get("WRITE_SIZE", 0.0)) * 1024,
    "tag": tag, "head": head, "kernels_sha": bench.kernel_sources_sha(), "photons": photons,
    "atomic_requests": job.get("TCC_EA0_ATOMIC_sum"),
    "per_kernel_GB": {k: {"read": 2 * v.get("FETCH_SIZE", 0.0) * 1024 / 1e9, "written": v.get("WRITE_SIZE", 0.0) * 1024 / 1e9}
                      for k, v in per_kernel.items()},
    "valu_wave_instructions": {k: v.get("SQ_INSTS_VALU") for k, v in per_kernel.items() if v.get("SQ_INSTS_VALU")},
    "note": "HBM bytes per job of one lane: (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the job's kernels, separate rocprofv3 "
            "--pmc passes (profiles/%s_%s_pmc_*.csv); head = the commit checked out when the summaries were collected, "
            "kernels_sha = bench.kernel_sources_sha() of the sources the passes ran on" % (tag, workload)}
json.dump(t, open(tf, "w"), indent=1)
print(json.dumps(t[key], indent=1))
