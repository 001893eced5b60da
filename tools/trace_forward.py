"""One forward of a profiled script, launch by launch, from a rocprofv3 kernel trace: python tools/trace_forward.py <kernel_trace.csv> <marker>
The forward is the span from the LAST launch whose name contains <marker> (the first kernel of a forward) to the end of the trace
(or to the next marker).  Prints the launches grouped by (kernel, grid) in order of first appearance, and the idle gaps between them."""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
i0 = marks[which]
i1 = marks[which + 1] if which != -1 and which + 1 < len(marks) else len(rows)
span = rows[i0:i1]
g = collections.OrderedDict()
gaps, last_end = 0, None
for r in span:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("sat::", "").replace("at::native::", "torch:")[:56]
    key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = g.setdefault(key, [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    if last_end is not None and s > last_end:
        d[2] += s - last_end
        gaps += s - last_end
    last_end = max(e, last_end or e)
tot = int(span[-1]["End_Timestamp"]) - int(span[0]["Start_Timestamp"])
print(f"forward: {len(span)} launches, {tot / 1e6:.2f} ms wall, {sum(v[1] for v in g.values()) / 1e6:.2f} ms of kernels, {gaps / 1e6:.2f} ms idle between them")
print("  n    avg us   total ms  idle-before ms  kernel  grid")
for (n, gx, gy, gz), (c, d, gp) in g.items():
    print(f"{c:4d} {d / c / 1e3:8.1f} {d / 1e6:9.3f} {gp / 1e6:9.3f}   {n}  {gx}x{gy}x{gz}")
