"""Print a window of a rocprofv3 kernel trace (csv) as a timeline: start offset, duration, gap to the previous kernel, name.
usage: trace_window.py <kernel_trace.csv> [first_row] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
a = sys.argv[2] if len(sys.argv) > 2 else str(max(len(rows) - 60, 0))
if ":" in a:                      # name:occurrence -> start at that occurrence of a kernel whose name contains `name`
    name, occ = a.split(":")
    a = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]][int(occ)]
a = int(a)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:a + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us  dur %8.1f  gap %8.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3,
                                                   r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
    prev_end = max(e, prev_end or 0)
