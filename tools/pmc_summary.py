"""Summarise rocprofv3 --pmc counter CSVs (one directory per pass) for one kernel into
profiles/hbm_traffic*.json (dev tool).
    python tools/pmc_summary.py <kernel-substring> <out.json> <pass-dir> [<pass-dir> ...]
Waves per solve: 1 for the wave kernels, 4 for solve_kernel_team<double, 4, ...> (nx = 512), 12 per trace for the
768-thread likelihood kernel; SQ_WAVES counts the waves the sampled shader engines saw, so per-solve figures are
ratios to it."""
import csv
import glob
import json
import os
import sys

kern, out = sys.argv[1], sys.argv[2]
vals = {}
for d in sys.argv[3:]:
    for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(fn)):
            if kern in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for c, per in acc.items():
            v = list(per.values())
            vals[c] = sum(v) / len(v)  # average per launch
fetch, write = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, --kernel-trace only) on "
                 "`python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --quick [--batch 2048 --opt kernel_variant=3 for solve_kernel_wave2; "
                 "--workload config5 for solve_kernel_team; --workload pt --chains 512 for loglik_kernel]` (one launch at a time on the device); kernel " + kern + ", average per launch",
       "solves_per_launch": 2048 if "wave2" in kern else (512 if ("team" in kern or "spike" in kern) else (3072 if "loglik" in kern else 1024)),
       "waves_per_solve": 4 if ("team" in kern or "spike" in kern) else (12 if "loglik" in kern else 1),
       "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of a wide coalesced read on gfx950 -> doubled; "
                     "WRITE_SIZE exact; both in KiB.  Reads here are small descriptor/time-grid loads, so the factor is an upper bound.",
       "solve_kernel_bytes_per_launch": (2 * fetch + write) * 1024 if fetch is not None and write is not None else None,
       "expected_compulsory_bytes_per_solve": (280 + 8 * 1001 + 4 + 32),
       "sq_counters_per_launch": {k: v for k, v in vals.items() if k.startswith("SQ_")}}
sq = res["sq_counters_per_launch"]
if "SQ_WAVES" in sq:
    w = sq["SQ_WAVES"] / res["waves_per_solve"]  # solves (traces) the sampled shader engines saw
    wc = sq.get("SQ_WAVE_CYCLES") or None
    frac = lambda k: (sq[k] / wc) if (wc and k in sq) else None
    res["derived"] = {"valu_insts_per_solve": sq.get("SQ_INSTS_VALU", 0) / w, "salu_insts_per_solve": sq.get("SQ_INSTS_SALU", 0) / w,
                      "lds_insts_per_solve": sq.get("SQ_INSTS_LDS", 0) / w,
                      "valu_active_fraction_of_wave_cycles": frac("SQ_ACTIVE_INST_VALU"),
                      "lds_active_fraction_of_wave_cycles": frac("SQ_ACTIVE_INST_LDS"),
                      "scalar_active_fraction_of_wave_cycles": frac("SQ_ACTIVE_INST_SCA"),
                      "parked_on_waitcnt_fraction_of_wave_cycles": frac("SQ_WAIT_ANY"),
                      "issue_stall_fraction_of_wave_cycles": frac("SQ_WAIT_INST_ANY"),
                      "occupancy_waves_per_simd": 2 if ("wave2" in kern or "spike" in kern) else (1 if "wave" in kern or "team" in kern or "spike" in kern else None),
                      "wave_cycles_per_solve(quad-cycles*4)": 4 * sq.get("SQ_WAVE_CYCLES", 0) / w / res["waves_per_solve"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
