"""group a rocprofv3 --kernel-trace csv by (kernel, grid size): launches and average duration — tells the stages of the
generator apart (the grid is the stage).  python tools/trace_by_grid.py <kernel_trace.csv> [forwards]"""
import collections
import csv
import re
import sys

per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("sat::", "")
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) if "Grid_Size_X" in r else int(r["Grid_Size"])
    per[(n, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
fw = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = sorted(per.items(), key=lambda kv: -sum(kv[1]))
for (n, grid), d in rows[:40]:
    print(f"{n[:64]:64s} grid {grid:>9d} launches/fwd {len(d) / fw:5.1f} avg_us {sum(d) / len(d) / 1e3:8.1f} ms/fwd {sum(d) / fw / 1e6:7.3f}")
