"""Absorbed one-token attention pool (k_apool_partial + merge, k_apool_dots + k_apool_bwd_apply + merge) alone at config 3's
shape (32 bags x 1024 keys x 512), HIP-event times: (a) on the same 64 MB of keys every call (Infinity-Cache resident),
(b) with 512 MB of other data streamed between calls (keys come from HBM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import ops
from mil_amd.segments import AttnSegs
dev = torch.device("cuda")
B, N, E, H, C = 32, 1024, 512, 8, 32
g = torch.Generator(device="cuda").manual_seed(1)
keys = torch.randn(B * N, E, device=dev, generator=g, requires_grad=True)
pe = torch.randn(N, E, device=dev, generator=g)
Qp = torch.randn(B, H, E, device=dev, generator=g, requires_grad=True) * 0.05
segs = AttnSegs.make([1] * B, [N] * B, dev)
big = torch.empty(128 * 1024 * 1024, device=dev)      # 512 MB


def timed(fn, iters=20, flush=False):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        if flush:
            big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def fwd():
    with torch.no_grad():
        return ops._AbsorbedPool.apply(keys, pe, Qp, segs, C)


pooled, kp = ops._AbsorbedPool.apply(keys, pe, Qp, segs, C)
dpooled = torch.randn_like(pooled)
dk = torch.randn_like(keys)


def bwd():
    torch.autograd.grad([pooled, kp], [keys, Qp], [dpooled, dk], retain_graph=True)


for flush in ([os.environ["FLUSH"] == "1"] if "FLUSH" in os.environ else (False, True)):
    print(f"flush={flush}: forward {timed(fwd, flush=flush):7.1f} us   backward {timed(bwd, flush=flush):7.1f} us")
