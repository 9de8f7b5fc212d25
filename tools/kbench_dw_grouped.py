#!/usr/bin/env python3
"""Times mil_linear_small_dw_grouped on the fusion step's token-side layer set (two two-way blocks + final attention + text
projection, 32 rows) and on subsets, 10 launches per replayed hipGraph.
    python tools/kbench_dw_grouped.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa: E402,F401
from mil_amd import _lib  # noqa: E402

dev = torch.device("cuda")
M = 32
BLOCK = [(512, 512), (512, 512), (256, 512), (512, 256), (2048, 512), (512, 2048), (256, 512), (512, 256)]   # (N, K)
ALL = BLOCK * 2 + [(256, 512), (512, 256), (512, 512)]


def build(layers):
    keep, arr = [], (_lib.SmallDwDesc * len(layers))()
    for d, (N, K) in zip(arr, layers):
        dy, y, x = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
        dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        keep += [dy, y, x, dW, db]
        d.dy, d.yv, d.x, d.dW, d.db = dy.data_ptr(), y.data_ptr(), x.data_ptr(), dW.data_ptr(), db.data_ptr()
        d.lddy, d.ldyv, d.ldx, d.lddw, d.act, d.M, d.N, d.K = N, N, K, K, 2, M, N, K
    return arr, keep


def timed(layers, n=10, reps=20):
    arr, keep = build(layers)
    st = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
    fn = lambda: _lib.check(_lib.lib().mil_linear_small_dw_grouped(arr, len(layers), st()), "dw")      # noqa: E731
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    par = sum(N * K for N, K in layers)
    return round(a.elapsed_time(b) / (n * reps) * 1e3, 1), par, sum(((N + 63) // 64) * ((K + 127) // 128) for N, K in layers)


for name, ls in (("all 19 layers", ALL), ("one block", BLOCK), ("lin1 + lin2", [(2048, 512), (512, 2048)]),
                 ("one 512 x 512", [(512, 512)]), ("4 x lin1", [(2048, 512)] * 4)):
    us, par, wgs = timed(ls)
    print(f"{name:16s} {us:7.1f} us   {par / 1e6:5.2f} M weights  {wgs:5d} workgroups  {par * 4 / us / 1e6:7.2f} TB/s written")
