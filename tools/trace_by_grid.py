"""usage: python tools/trace_by_grid.py <rocprofv3 kernel_trace.csv>  -- average / median / minimum duration per (kernel, grid): tells the two launches of one
kernel apart (the H-side and the W-side product, the H and the W update), which rocprofv3's --stats summary adds up."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else None]:
    v2 = sorted(v)
    print(f"{k[0]:62s} grid {k[1]:>7s} x {k[2]:>3s} x {k[3]:>2s} (block {k[4]:>4s}) calls {len(v):5d}  avg {sum(v) / len(v) / 1e3:8.2f}  median {v2[len(v2) // 2] / 1e3:8.2f}  min {v2[0] / 1e3:8.2f} us")
