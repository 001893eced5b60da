"""What the GPU does during the timed steps of the headline bench, from a rocprofv3 kernel trace: how many kernels are in flight
over time, how long nothing runs, and per kernel family its in-bench duration.  python tools/trace_timeline.py <kernel_trace.csv> [forwards in the window, default 16]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
names = collections.defaultdict(lambda: [0, 0.0])
# the timed steps: the densest run of generator forwards (one mrf16 / convpost launch each) — the roofline legs that follow the
# steps in bench.py run one forward at a time with host synchronisation between them and would dilute a whole-run window
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "convpost" in r["Kernel_Name"])
n_fw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
# ... and those legs are generator-only: a window of the steps also holds one F0 extraction per forward
import bisect
f0 = sorted(int(r["Start_Timestamp"]) for r in rows if "yaapt_nlfer" in r["Kernel_Name"])
cand = [i for i in range(len(marks) - n_fw) if bisect.bisect(f0, marks[i + n_fw]) - bisect.bisect(f0, marks[i]) >= n_fw - 2]
i0 = min(cand or range(len(marks) - n_fw), key=lambda i: marks[i + n_fw] - marks[i])
lo, hi = marks[i0], marks[i0 + n_fw]
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= lo or s >= hi:
        continue
    s, e = max(s, lo), min(e, hi)
    ev.append((s, 1))
    ev.append((e, -1))
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("sat::", "")[:48]
    names[n][0] += 1
    names[n][1] += e - s
ev.sort()
depth, last, hist = 0, lo, collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
hist[depth] += hi - last
tot = hi - lo
print(f"densest window {tot / 1e6:.1f} ms = {n_fw} generator forwards ({tot / 1e6 / n_fw:.2f} ms each); kernels in flight -> share of the time:")
for k in sorted(hist):
    print(f"  {k:2d}: {100 * hist[k] / tot:5.1f} %")
print(f"mean kernels in flight {sum(k * v for k, v in hist.items()) / tot:.2f}")
print("kernel families by in-flight time (share of the window x kernels):")
for n, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {100 * d / tot:6.1f} %  {c:5d} x {d / c / 1e3:8.1f} us  {n}")
