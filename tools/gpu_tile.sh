#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmc_tile -- python3 tools/tile_capture.py > gpurun_out/pmc_tile.log 2>&1
cat gpurun_out/pmc_tile.log | grep captured
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob("gpurun_out/pmc_tile/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "walk_kernel" in r["Kernel_Name"]:
        print(r["Dispatch_Id"], r["Counter_Name"], r["Counter_Value"])
PY
