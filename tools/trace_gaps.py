"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace (one stream): where a steady-state iteration idles.
usage: trace_gaps.py DIR_OR_CSV [LAST_N=120]   prints name:duration(+gap before) for the last N kernels and the idle time per kernel name."""
import csv, os, sys, collections
p = sys.argv[1]
if os.path.isdir(p):
    p = [os.path.join(r, f) for r, _, fs in os.walk(p) for f in fs if f.endswith("kernel_trace.csv")][0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
short = lambda s: s.split("(")[0].replace("void ", "").replace("nmfamd::", "").replace("(anonymous namespace)::", "")[:28]
prev = None
out, idle = [], collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    out.append(f"{short(r['Kernel_Name'])}:{(e - s) / 1e3:.1f}(+{gap:.1f})")
    if 0 < gap < 200:
        idle[short(r["Kernel_Name"])][0] += gap; idle[short(r["Kernel_Name"])][1] += 1
    prev = e
print(" ".join(out))
print("idle time in front of (us total, launches with a gap):", {k: (round(v[0], 1), v[1]) for k, v in sorted(idle.items(), key=lambda kv: -kv[1][0])})
