"""Copy the rocprof summaries of the last tools/gpu_profile.sh run into profiles/<tag>_* and
refresh profiles/traffic.json (PMC HBM bytes per launch of the bench kernel)."""
import csv, glob, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "f64walk_f64"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
def first(pat):
    f = sorted(glob.glob(os.path.join(G, pat)), key=os.path.getmtime)   # newest: gpurun_out keeps older runs
    return f[-1] if f else None
shutil.copy(first("prof_trace/*/*_kernel_stats.csv"), os.path.join(P, tag + "_kernel_stats.csv"))
vals = collections.defaultdict(list)
for d in ("prof_fetch", "prof_write", "prof_sq"):
    f = first(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(P, "%s_pmc_%s.csv" % (tag, d.split("_")[1])))
    for r in csv.DictReader(open(f)):
        if "walk_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
line = [l for l in open(os.path.join(G, "bench_n1.log")) if l.startswith("{")][-1]
open(os.path.join(P, tag + "_bench_n1.json"), "w").write(line)
tf = os.path.join(P, "traffic.json")
t = json.load(open(tf)) if os.path.exists(tf) else {}
# MI355X_MICROARCH "HBM": FETCH_SIZE under-reports wide reads by 2x on gfx950 (doubled here; it is ~0 anyway);
# WRITE_SIZE is in KiB.  bytes per launch:
t[key] = (2 * mean.get("FETCH_SIZE", 0.0) + mean.get("WRITE_SIZE", 0.0)) * 1024
t[key + "_atomic_requests"] = mean.get("TCC_EA0_ATOMIC_sum")
t["_note"] = ("bytes per launch of the bench's walk_kernel (C2, 1e7 photons): (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate "
              "rocprofv3 --pmc passes (profiles/%s_pmc_*.csv)" % tag)
json.dump(t, open(tf, "w"), indent=1)
print(json.dumps(mean, indent=1)); print(line[:400])
