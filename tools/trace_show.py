"""Kernel timeline of a rocprofv3 --kernel-trace run: the last `n` kernels (memsets left out), times in ms relative to the first shown.
    python tools/trace_show.py <dir> [n]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = [r for r in csv.DictReader(open(f)) if "fillBuffer" not in r["Kernel_Name"] and "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    if "walk_kernel" in name:
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")      # R, GEOM, TABLE, TALLY, CAPTURE[, PHASE]
        short = "walk" + {"0": "", "1": " main", "2": " TAIL"}.get(args[5] if len(args) > 5 else "0", "?")
    else:
        short = name[name.index("k_log_"):].split("(")[0].split("<")[0] if "k_log_" in name else name[:24]
    print("%-16s q%-3s %9.3f -> %9.3f  (%7.3f ms)" % (short, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6,
                                                     (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
