"""Cost of one dependent kernel node in a replayed hipGraph (and eagerly): N x k_counter_add (1 thread) back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops

dev = torch.device("cuda")
c = torch.zeros(1, device=dev, dtype=torch.int32)
a = torch.randn(32, 512, device=dev)
W = torch.randn(512, 512, device=dev)
b = torch.randn(512, device=dev)
for name, fn in (("counter_add", lambda: ops.counter_add(c, 1)),
                 ("small_fwd 32x512x512", lambda: ops.linear_small_fwd(a, W, b, 0)),
                 ("torch add [32,512]", lambda: a + a)):
    for N in (100, 400):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(N):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 20 / N * 1e6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N * 5):
            fn()
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / (N * 5) * 1e6
        print(f"{name:24s} N={N:4d}: graph {tg:6.2f} us/node   eager {te:6.2f} us/launch")
