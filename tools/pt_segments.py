"""Replica-exchange segments of the PT loop from a rocprofv3 kernel trace (dev tool): per segment (the launches between
two exchanges), when each queue's first solve starts and last decision ends, and the solve durations.
    python tools/pt_segments.py <..._kernel_trace.csv> [gap_ms that separates segments, default 0.8]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "solve_kernel" in n or "loglik" in n:
        ev.append((int(r["Start_Timestamp"]) / 1e6, int(r["End_Timestamp"]) / 1e6, "S" if "solve_kernel" in n else "L", r["Queue_Id"]))
ev.sort()
qs = sorted({e[3] for e in ev if e[2] == "S"}, key=lambda q: -sum(1 for e in ev if e[3] == q))[:4]
ev = [e for e in ev if e[3] in qs]
# a segment ends where NO solve is running for `gap` ms
segs, cur, busy_until = [], [], None
for e in ev:
    if busy_until is not None and e[0] - busy_until > gap and e[2] == "S":
        segs.append(cur)
        cur = []
    cur.append(e)
    busy_until = e[1] if busy_until is None else max(busy_until, e[1])
segs.append(cur)
prev_end = None
for s in segs[-8:]:
    t0 = min(e[0] for e in s)
    t1 = max(e[1] for e in s)
    line = f"segment {t1 - t0:7.2f} ms" + (f" (idle before {t0 - prev_end:5.2f})" if prev_end else "")
    for q in qs:
        sq = [e for e in s if e[3] == q and e[2] == "S"]
        if sq:
            line += f" | q{q}: start +{sq[0][0] - t0:4.2f} end +{max(e[1] for e in s if e[3] == q) - t0:5.2f} solves " + " ".join(f"{b - a:.2f}" for a, b, *_ in sq)
    print(line)
    prev_end = t1
