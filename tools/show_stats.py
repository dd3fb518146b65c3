import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print(f"{name:44s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
