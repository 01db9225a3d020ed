"""Diagnostic: gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (usage: trace_gaps.py <kernel_trace.csv> [min_gap_us])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
out = []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-28:]
    if prev is not None:
        gap = (s - prev[1]) / 1000.0
        if gap > thr:
            out.append((i, gap, prev[2], name))
    prev = (s, e, name)
print("kernels", len(rows), "gaps >", thr, "us:", len(out))
for i, gap, a, b in out[-60:]:
    print(f"  #{i:6d} gap {gap:9.1f} us  {a} -> {b}")
