"""Per-step kernel totals of a replayed bucket step from a rocprofv3 kernel trace: the dispatches between consecutive launches
of a marker kernel (default k_build_fusion_segs, the first launch of a RaggedFusionStepper step), averaged over steps 20 - 40.
usage: step_kernels.py <kernel_trace.csv> [marker] [top]"""
import collections
import csv
import sys

rows = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:64], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_build_fusion_segs"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
idx = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
agg = collections.defaultdict(lambda: [0, 0.0])
n = 0
for a, b in zip(idx[20:40], idx[21:41]):
    n += 1
    for name, s, e in rows[a:b]:
        agg[name][0] += 1
        agg[name][1] += (e - s) / 1e3
print("kernels per step %.0f, kernel time per step %.1f us, span %.1f us" % (
    sum(v[0] for v in agg.values()) / n, sum(v[1] for v in agg.values()) / n, (rows[idx[40]][1] - rows[idx[20]][1]) / 1e3 / n))
for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.1f us  x%5.1f  %s" % (t / n, c / n, name))
