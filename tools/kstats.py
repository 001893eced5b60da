"""print the top rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.1f}%  {r['Name'][:70]}")
