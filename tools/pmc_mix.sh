#!/bin/bash
# tools/pmc_mix.sh TAG -- floating-point instruction mix of the two-per-SIMD wave kernel (rocprofv3 --pmc, each group in its
# own pass, --kernel-trace only) -> gpurun_out/hbm_traffic_TAG_mix_wave2.json (copied to profiles/ by collect_profiles.sh);
# and the issue cost of the instruction classes on this device (tools/mb/rcp_rate.hip) -> gpurun_out/TAG_issue_cost.txt
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
O=gpurun_out
B2="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --quick --batch 2048 --opt kernel_variant=3"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $O/pmc${T}_mix64 -- $B2 > $O/pmc${T}_mix1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $O/pmc${T}_mix32 -- $B2 > $O/pmc${T}_mix2.log 2>&1
python - <<PY
import csv, glob, collections, json
out = {"source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 | ..._F32 (two passes, --kernel-trace only) on "
                 "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --quick --batch 2048 --opt kernel_variant=3; solve_kernel_wave2, wave-instructions per solve (= per wave)"}
for d in ("$O/pmc${T}_mix64", "$O/pmc${T}_mix32"):
    acc = collections.defaultdict(float)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "solve_kernel_wave2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if k != "SQ_WAVES":
            out.setdefault("per_solve", {})[k] = v / acc["SQ_WAVES"]
p = out["per_solve"]
f64 = sum(p[k] for k in p if k.endswith("_F64") and "TRANS" not in k)
f32 = sum(p[k] for k in p if k.endswith("_F32") and "TRANS" not in k)
out["classes"] = {"fp64": f64, "fp64_transcendental": p["SQ_INSTS_VALU_TRANS_F64"], "fp32": f32,
                  "fp32_transcendental": p["SQ_INSTS_VALU_TRANS_F32"],
                  "other (moves, conversions, selects, DPP, integer, lane reads)": p["SQ_INSTS_VALU"] - f64 - f32 - p["SQ_INSTS_VALU_TRANS_F64"] - p["SQ_INSTS_VALU_TRANS_F32"]}
json.dump(out, open("$O/hbm_traffic_${T}_mix_wave2.json", "w"), indent=1)
print(json.dumps(out["classes"], indent=1))
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $O/rcp_rate tools/mb/rcp_rate.hip > $O/rcp_rate_build.log 2>&1
for w in 1 2 4; do echo "== $w wave(s) per SIMD"; timeout -k 5 60 $O/rcp_rate $w; done > $O/${T}_issue_cost.txt 2>&1
cat $O/${T}_issue_cost.txt | cut -c1-120
