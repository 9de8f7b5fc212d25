"""The frozen text tower's products at M = 770 rows (one bag x 10 prompts x 77 tokens: the learnable-prompt step of the one-bag
regime) - forward y = x W^T and the input gradient dx = dy W, per launch inside a hipGraph, against the 157 TF fp32 peak."""
import sys, os, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import ops
dev = torch.device("cuda")


def t(fn, n=20, reps=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


M = int(os.environ.get("M", "770"))
for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    W = (torch.randn(N, K, device=dev) / K ** 0.5)
    b = torch.zeros(N, device=dev)
    with torch.no_grad():
        f = t(lambda: ops.linear_act(x, W, b))
    dy = torch.randn(M, N, device=dev)
    bw = t(lambda: ops.gemm(dy, 0, W, 1, M, K, N))          # what _LinearAct.backward launches for dx (frozen weight)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: fwd {f:.1f} us = {fl / f / 1e6:.0f} TF   dx {bw:.1f} us = {fl / bw / 1e6:.0f} TF")
