#!/bin/bash
# GPU box: rocprofv3 kernel stats + SQ / memory PMC passes of one configuration run directly through the C ABI
# (tools/lane_trace.py run 1 <case>: 3 launches, one lane).   bash tools/gpu_profile_case.sh c4
set -o pipefail
CASE=${1:-c4}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/case_trace_$CASE -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/case_trace_$CASE.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/case_sq_$CASE -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/case_sq_$CASE.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/case_fetch_$CASE -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/case_fetch_$CASE.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/case_write_$CASE -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/case_write_$CASE.log 2>&1
echo rc=$?
cat gpurun_out/case_trace_$CASE.log | grep launch
python3 - $CASE <<'PY'
import csv, glob, os, sys
case = sys.argv[1]
for d in ("sq", "fetch", "write"):
    fs = sorted(glob.glob("gpurun_out/case_%s_%s/**/*counter_collection.csv" % (d, case), recursive=True), key=os.path.getmtime)
    if not fs: continue
    last = {}
    for r in csv.DictReader(open(fs[-1])):
        kn = r["Kernel_Name"]
        name = "walk" if "walk_kernel" in kn else (kn[kn.index("k_log_"):].split("(")[0] if "k_log_" in kn else None)
        if name: last.setdefault(name, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in last.items():
        print(d, k, {a: ("%.4g" % b) for a, b in v.items()})
PY
