"""Does a bandwidth-bound launch (Adam over the fusion model's 9.5 M parameters) hide under an MFMA-bound one (fc_pathology's
32768 x 768 x 512 product) when a captured hipGraph forks it onto a second stream?  Serial vs forked, eager and replayed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mil_amd  # noqa
from mil_amd import ops

dev = torch.device("cuda")
R, K, N = 32768, 768, 512
x = torch.randn(R, K, device=dev)
W = torch.randn(N, K, device=dev) * 0.02
b = torch.zeros(N, device=dev)
n = 9_500_000
P, G, M1, M2 = (torch.zeros(n, device=dev) for _ in range(4))
G.normal_()
ctr = torch.zeros(1, device=dev, dtype=torch.int32)
lr = torch.full((1,), 1e-5, device=dev)
side = torch.cuda.Stream()


def gemm():
    with torch.no_grad():
        return ops.linear_act(x, W, b, "tanh")


def adam():
    ops.adam_step_dev(P, G, M1, M2, ctr, lr)


def serial():
    gemm(); adam()


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        adam()
    gemm()
    main.wait_stream(side)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        fn()
    return g.replay


for _ in range(100):
    gemm()
print("gemm alone      %.4f ms" % timeit(gemm))
print("adam alone      %.4f ms" % timeit(adam))
print("serial eager    %.4f ms" % timeit(serial))
print("forked eager    %.4f ms" % timeit(forked))
def rep(fn, k=10):
    def f():
        for _ in range(k):
            fn()
    return f


def gemm_only():
    gemm()


print("10 x gemm graph   %.4f ms each" % (timeit(graphed(rep(gemm_only))) / 10))
print("10 x adam graph   %.4f ms each" % (timeit(graphed(rep(adam))) / 10))
print("10 x serial graph %.4f ms each" % (timeit(graphed(rep(serial))) / 10))
print("10 x forked graph %.4f ms each" % (timeit(graphed(rep(forked))) / 10))
