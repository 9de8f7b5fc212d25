"""A/B of the bf16 pool partial pass at config 5 (32 bags x 4096 x 1024): this build against every tools/variants/*.so
(e.g. the previous commit's library copied there as libprev.so), HIP events around 30 launches, three interleaved rounds."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import _lib, ops, synthetic as syn
from mil_amd.bags import BagLayout

dev = torch.device("cuda")
B, N, L = 32, 4096, 1024
x16 = ops.cast_bf16(torch.randn((B * N, L), device=dev))
scores = torch.randn(B * N, device=dev)
lay = BagLayout.uniform(B, N, dev)
Wf = torch.randn((2, L), device=dev)


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def load(path):
    h = ctypes.CDLL(path)
    for name, (res, a) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, a
    return h


libs = {"main": _lib.lib()}
for q in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so"))):
    libs[os.path.basename(q)] = load(q)
ref = None
for rnd in range(3):
    for name, h in libs.items():
        _lib._lib = h
        t = timed(lambda: ops.attn_pool_partial_h_bf16(x16, scores, lay, Wf))
        part, hrow = ops.attn_pool_partial_h_bf16(x16, scores, lay, Wf)
        if ref is None:
            ref = (part.clone(), hrow.clone())
        print(f"round {rnd} {name:24s} pool_partial_h_bf16 {t:7.1f} us  = {B * N * L * 2 / t / 1e6:5.2f} TB/s   "
              f"bit-equal to main: {bool(torch.equal(part, ref[0]) and torch.equal(hrow, ref[1]))}")
_lib._lib = libs["main"]
