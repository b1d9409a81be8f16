import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=float(sys.argv[2]) if len(sys.argv)>2 else 1
tot=0
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 18]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:60]:60s} calls={r['Calls']:>4s} ms/step={int(r['TotalDurationNs'])/1e6/steps:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}%")
print("total ms/step", sum(int(r['TotalDurationNs']) for r in rows)/1e6/steps)
