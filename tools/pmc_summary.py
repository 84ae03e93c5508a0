"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv (last job's dispatches) and the derived SQ figures.
    python tools/pmc_summary.py <csv> [steps_per_job]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
by = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
last_walk = max(int(r["Dispatch_Id"]) for r in rows if "walk_kernel" in r["Kernel_Name"] and ", 2>" not in r["Kernel_Name"])
for r in rows:
    if int(r["Dispatch_Id"]) < last_walk: continue
    kn = r["Kernel_Name"]
    if not ("walk_kernel" in kn or "k_log_" in kn): continue
    name = kn[kn.index("k_log_"):].split("(")[0] if "k_log_" in kn else kn.split("(")[0].split("::")[-1]
    by[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name].add(r["Dispatch_Id"])
for k, v in by.items():
    s = "%-44s x%d " % (k[:44], len(cnt[k]))
    if "SQ_INSTS_VALU" in v:
        s += "VALU %.4g" % v["SQ_INSTS_VALU"]
        if steps and "walk" in k: s += " (%.0f per photon-step)" % (v["SQ_INSTS_VALU"] * 64 / steps)
        if v.get("SQ_ACTIVE_INST_VALU"): s += " lane util %.1f%%" % (100 * v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]))
        if v.get("SQ_WAVE_CYCLES"): s += " wait_any %.1f%% busy/wave %.3f" % (100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"])
    else:
        s += " ".join("%s %.4g" % (a, b) for a, b in v.items())
    print(s)
