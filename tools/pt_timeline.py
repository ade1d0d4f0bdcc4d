"""Launch cycles of the PT loop from a rocprofv3 kernel trace (dev tool): per HIP queue, the solve kernel's duration, the
likelihood kernel's, and the gap from the end of one solve to the start of the next on the same queue.
    python tools/pt_timeline.py <..._kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
byq = {}
for r in rows:
    n = r["Kernel_Name"]
    kind = "solve" if "solve_kernel" in n else ("loglik" if "loglik" in n else None)
    if kind:
        byq.setdefault(r["Queue_Id"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
for q, ev in byq.items():
    ev.sort()
    sol = [(a, b) for a, b, k in ev if k == "solve"]
    lik = [(b - a) / 1e6 for a, b, k in ev if k == "loglik"]
    dur = np.array([(b - a) / 1e6 for a, b in sol])
    gap = np.array([(sol[i + 1][0] - sol[i][1]) / 1e6 for i in range(len(sol) - 1)])
    if len(sol) < 4:
        continue
    print(f"queue {q}: {len(sol)} solve launches; solve ms median {np.median(dur):.3f} mean {dur.mean():.3f} max {dur.max():.3f}; "
          f"loglik median {np.median(lik):.3f}; solve-end -> next solve-start median {np.median(gap):.3f} mean {gap.mean():.3f} max {gap.max():.3f}")
    print("   solve ms:", " ".join(f"{d:.2f}" for d in dur[-24:]))
    print("   gap ms:  ", " ".join(f"{d:.2f}" for d in gap[-23:]))
