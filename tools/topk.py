import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:int(sys.argv[2])]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f}us tot={float(r['TotalDurationNs'])/1e6:8.2f}ms {float(r['Percentage']):5.1f}%")
