"""Diagnostic: the last N kernels of a rocprofv3 --kernel-trace CSV as a timeline (start / duration / gap to the previous
end, in us; overlapping kernels show a negative gap).  Usage: trace_timeline.py <kernel_trace.csv> [N] [skip_from_end] [name: the window ends at its last launch]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 4:                                    # the window ENDS at the last kernel whose name contains argv[4]
    last = max(i for i, r in enumerate(rows) if sys.argv[4] in r["Kernel_Name"])
    rows = rows[max(0, last + 1 - n - skip):last + 1 - skip]
else:
    rows = rows[len(rows) - n - skip:len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:28]
    gap = "" if prev_end is None else f"{(s - prev_end) / 1000.0:8.1f}"
    print(f"{(s - t0) / 1000.0:10.1f} {(e - s) / 1000.0:8.1f} {gap:>8}  q{r.get('Queue_Id', '?'):>3} {name}")
    prev_end = max(prev_end, e) if prev_end is not None else e
