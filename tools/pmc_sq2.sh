#!/bin/bash
# tools/pmc_sq2.sh TAG VARIANT BATCH -- instruction-fetch / LDS-latency / VALU-mix counter passes (run on the GPU box)
T=$1; V=$2; B=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
CMD="python3 tools/variant_bench.py --variants $V --batches $B --reps 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${T}_c -- $CMD > $O/pmc_${T}_c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $O/pmc_${T}_d -- $CMD > $O/pmc_${T}_d.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SMEM SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $O/pmc_${T}_e -- $CMD > $O/pmc_${T}_e.log 2>&1
python3 - <<PY
import csv, glob
vals = {}
for d in ("$O/pmc_${T}_c", "$O/pmc_${T}_d", "$O/pmc_${T}_e"):
    for fn in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = {}
        for row in csv.DictReader(open(fn)):
            if "solve_kernel" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for c, per in acc.items():
            v = list(per.values())
            vals[c] = sum(v) / len(v)
w = $B
print("variant $V batch $B: per solve")
for k in sorted(vals):
    print(f"  {k:28s} {vals[k] / w:14.0f}")
PY
