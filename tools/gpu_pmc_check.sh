#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
LT_DIAG_NO_TALLY=1 rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmc_notally -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_notally.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmc_small -- python3 bench.py --steps 1 --warmup 0 --photons 1000000 --no-cpu-baseline > gpurun_out/pmc_small.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_ATOMIC_sum --output-format csv -d gpurun_out/pmc_more -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_more.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob
for d in ("pmc_notally","pmc_small","pmc_more"):
    for f in glob.glob("gpurun_out/%s/*/*_counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "walk_kernel" in r["Kernel_Name"]:
                print(d, r["Counter_Name"], r["Counter_Value"])
PY
tail -3 gpurun_out/pmc_more.log
