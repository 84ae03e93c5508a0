"""Copy the rocprof summaries of the last tools/gpu_profile.sh run into profiles/<tag>_* and refresh
profiles/traffic.json.  Counters are taken from the LAST job of each PMC pass: the last walk_kernel dispatch and
the k_log_* dispatches that follow it (one complete C2 job: walk + hist + scan + 2 partition passes + reduce)."""
import csv, glob, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "f64walk_f64_log"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
def newest(pat):
    f = sorted(glob.glob(os.path.join(G, pat)), key=os.path.getmtime)   # gpurun_out keeps older runs
    return f[-1] if f else None
shutil.copy(newest("prof_trace/*/*_kernel_stats.csv"), os.path.join(P, tag + "_kernel_stats.csv"))
job = collections.defaultdict(float)                                    # counter -> sum over the job's kernels
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("prof_fetch", "prof_write", "prof_sq"):
    f = newest(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(P, "%s_pmc_%s.csv" % (tag, d.split("_")[1])))
    rows = [r for r in csv.DictReader(open(f)) if "walk_kernel" in r["Kernel_Name"] or "k_log_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last_walk = max(int(r["Dispatch_Id"]) for r in rows if "walk_kernel" in r["Kernel_Name"])
    for r in rows:
        if int(r["Dispatch_Id"]) >= last_walk:
            kn = r["Kernel_Name"]
            name = kn[kn.index("k_log_"):].split("(")[0] if "k_log_" in kn else kn.split("(")[0].split("::")[-1]
            job[r["Counter_Name"]] += float(r["Counter_Value"])
            per_kernel[name][r["Counter_Name"]] += float(r["Counter_Value"])
line = [l for l in open(os.path.join(G, "bench_n1.log")) if l.startswith("{")][-1]
open(os.path.join(P, tag + "_bench_n1.json"), "w").write(line)
tf = os.path.join(P, "traffic.json")
t = json.load(open(tf)) if os.path.exists(tf) else {}
# MI355X_MICROARCH "HBM": on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads (doubled here);
# WRITE_SIZE is exact for 16-B-per-lane streaming stores and float atomics; both in KiB.
t[key] = (2 * job.get("FETCH_SIZE", 0.0) + job.get("WRITE_SIZE", 0.0)) * 1024
t[key + "_atomic_requests"] = job.get("TCC_EA0_ATOMIC_sum")
t[key + "_per_kernel_GB"] = {k: {"read": 2 * v.get("FETCH_SIZE", 0.0) * 1024 / 1e9, "written": v.get("WRITE_SIZE", 0.0) * 1024 / 1e9}
                             for k, v in per_kernel.items()}
t["_note"] = ("HBM bytes per C2 job (1e7 photons): (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the job's kernels, from "
              "separate rocprofv3 --pmc passes (profiles/%s_pmc_*.csv)" % tag)
json.dump(t, open(tf, "w"), indent=1)
print(json.dumps({k: v for k, v in job.items()}, indent=1)); print(line[:300])
for kn, d in per_kernel.items():
    print("%-28s" % kn, {k: round(v, 1) for k, v in d.items() if k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum", "SQ_INSTS_VALU", "SQ_WAVES")})
