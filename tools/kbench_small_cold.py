"""Token-side linear layer (32 x 512 -> 512) in a dependent chain inside one hipGraph: the same weight every layer (L2-warm) vs
48 distinct weights in turn (48 MB: every layer's weight comes from Infinity Cache / HBM, as in the model's step)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import ops
dev = torch.device("cuda")
M, N, K, NL = 32, 512, 512, 48
Ws = [torch.randn(N, K, device=dev) / K ** 0.5 for _ in range(NL)]
b = torch.zeros(N, device=dev)
x0 = torch.randn(M, K, device=dev)


def chain(distinct):
    x = x0
    for i in range(NL):
        x = ops.linear_small_fwd(x, Ws[i if distinct else 0], b, 1)
    return x


def t(fn, reps=30):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / (NL * reps) * 1e3


print("same weight    %.2f us / layer" % t(lambda: chain(False)))
print("48 weights     %.2f us / layer" % t(lambda: chain(True)))
