"""Print the kernel sequence of ONE step from a rocprofv3 kernel trace (the dispatches between two consecutive launches of a
marker kernel), with durations: where in the step do the torch-owned launches sit?  usage: trace_seq.py <trace.csv> <marker>"""
import csv, sys
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:70], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
marker = sys.argv[2]
idx = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
a, b = idx[-3], idx[-2]
prev_end = rows[a][1]
for n, s, e in rows[a:b]:
    print(f"{(e - s) / 1e3:8.2f} us  gap {(s - prev_end) / 1e3:6.2f}  {n}")
    prev_end = e
print("step span %.1f us, kernels %d" % ((rows[b][1] - rows[a][1]) / 1e3, b - a))
