import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:70]
    d[k][0] += 1; d[k][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:72s} calls={v[0]:5d} avg={v[1]/v[0]:9.2f}us pct={100*v[1]/tot:5.2f}")
