#!/bin/bash
# PMC passes over one-lane C2 launches: which resource holds the log kernels.  Output: gpurun_out/pmc_*/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CASE=${1:-c2}
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/pmc_sq.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_lds -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/pmc_lds.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/pmc_write -- python3 tools/lane_trace.py run 1 $CASE > gpurun_out/pmc_write.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections, os
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    fs = sorted(glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True), key=os.path.getmtime)
    if not fs: continue
    rows = list(csv.DictReader(open(fs[-1])))
    last = {}
    for r in rows:   # keep the last dispatch of every kernel
        kn = r["Kernel_Name"]
        name = "walk" if "walk_kernel" in kn else (kn[kn.index("k_log_"):].split("(")[0] if "k_log_" in kn else None)
        if name: last.setdefault(name, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in last.items():
        print(d, k, {a: ("%.4g" % b) for a, b in v.items()})
PY
