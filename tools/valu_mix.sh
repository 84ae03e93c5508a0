# Dynamic VALU instruction mix of one C2 job (one lane, kernels one after another): rocprofv3 PMC passes over the per-type counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="bench.py --workload ${1:-c2} --inflight 1 --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline --no-alone --extras none"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/mix_$i -- python3 $A > gpurun_out/mix_$i.log 2>&1 || { echo "set $i failed: $set"; tail -3 gpurun_out/mix_$i.log; continue; }
done
python3 - <<'PY'
import csv, glob, os, collections
tot = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/mix_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "walk_kernel" in r["Kernel_Name"] and ", 1>(" in r["Kernel_Name"]]
    if not rows: continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last: tot["bulk walk"][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in tot.items():
    base = v.get("SQ_INSTS_VALU", 1.0)
    print(k, "SQ_INSTS_VALU = %.4g" % base)
    for c, x in sorted(v.items(), key=lambda t: -t[1]):
        print("   %-28s %.4g  (%.1f %% of VALU)" % (c, x, 100 * x / base))
PY
rm -rf gpurun_out/mix_1 gpurun_out/mix_2 gpurun_out/mix_3 gpurun_out/mix_4
