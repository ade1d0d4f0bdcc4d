#!/bin/bash
# tools/pmc_sq.sh TAG VARIANT BATCH -- SQ counter passes of one solver backend (run on the GPU box via gpurun)
T=$1; V=$2; B=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
CMD="python3 tools/variant_bench.py --variants $V --batches $B --reps 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_${T}_a -- $CMD > $O/pmc_${T}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/pmc_${T}_b -- $CMD > $O/pmc_${T}_b.log 2>&1
python3 - <<PY
import csv, glob
vals = {}
for d in ("$O/pmc_${T}_a", "$O/pmc_${T}_b"):
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
    print(f"  {k:24s} {vals[k] / w:14.0f}")
PY
