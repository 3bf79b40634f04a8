"""Per-(kernel, grid) averages of a rocprofv3 kernel trace: the two products of an iteration are one template instance and differ by their grids.
usage: trace_by_grid.py DIR_OR_CSV [min_calls]"""
import csv, os, sys, collections
p = sys.argv[1]
if os.path.isdir(p):
    p = [os.path.join(r, f) for r, _, fs in os.walk(p) for f in fs if f.endswith("kernel_trace.csv")][0]
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = collections.defaultdict(list)
for r in csv.DictReader(open(p)):
    d[(r["Kernel_Name"].split("(")[0][:64], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < mc: continue
    v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v      # trimmed mean: warm-up launches out
    print(f"{k[0]:66s} wgs={k[1]:5d} y={k[2]} calls={len(v):5d} avg={sum(v)/len(v):8.2f} trimmed={sum(v2)/len(v2):8.2f} us")
    tot += sum(v2) / len(v2)
print(f"sum of trimmed averages: {tot:.1f} us")
