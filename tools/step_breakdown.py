#!/usr/bin/env python3
"""Per-kernel launches / time of ONE replayed step from a rocprofv3 kernel trace: the dispatches between consecutive launches
of a marker kernel (one per step), averaged over the last N steps.
    python tools/step_breakdown.py gpurun_out/cfg3prof k_gemm_nt2 [15]"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:70]


def main():
    d, marker = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    import os
    f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r[1])
    idx = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
    use = list(zip(idx[-steps - 1:-1], idx[-steps:]))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for a, b in use:
        for n, s0, e0 in rows[a:b]:
            agg[n][0] += 1
            agg[n][1] += (e0 - s0) / 1e3
    n = len(use)
    span = (rows[use[-1][1]][1] - rows[use[0][0]][1]) / 1e3 / n
    print(f"launches/step {sum(v[0] for v in agg.values()) / n:.1f}  kernel time/step {sum(v[1] for v in agg.values()) / n:.1f} us  span {span:.1f} us")
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:72s} {c / n:6.2f} {t / n:9.2f} {t / c:8.2f}")


if __name__ == "__main__":
    main()
