"""Time the bf16 weight-gradient kernel (k_gate_bwd_dw_bf16 + reduce) alone at config 5; also every library under
tools/variants/ (ablation builds: results of those are not meaningful, only their times)."""
import os, sys, glob, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops, _lib

def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda")
B, N, L = 32, 4096, 1024
R = B * N
torch.manual_seed(0)
x16 = ops.cast_bf16(torch.randn((R, L), device=dev))
gates = torch.rand((R, 384), device=dev).to(torch.bfloat16)
ds = torch.randn((R,), device=dev) * 1e-3
w = torch.randn((192,), device=dev)
outs = [torch.empty((192, L), device=dev), torch.empty(192, device=dev), torch.empty((192, L), device=dev), torch.empty(192, device=dev),
        torch.empty(192, device=dev), torch.empty(1, device=dev)]
flops = 4.0 * R * L * 192
def run(): ops.gate_bwd_params_bf16(x16, gates, ds, w, *outs)
def load(path):
    h = ctypes.CDLL(path)
    for name, (res, a) in _lib.SIGNATURES.items():
        fn = getattr(h, name); fn.restype = res; fn.argtypes = a
    return h
_lib.lib()
t = timed(run); ref = [o.clone() for o in outs]
print(f"main                          dW bf16 + reduce {t:7.1f} us  {flops / t / 1e6:7.1f} TF")
for q in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so"))):
    keep = _lib._lib
    _lib._lib = load(q)
    t = timed(run)
    err = max(float((a - b).abs().max()) for a, b in zip(outs, ref))
    _lib._lib = keep
    print(f"variant {os.path.basename(q):28s} {t:7.1f} us  {flops / t / 1e6:7.1f} TF   max|d| vs main {err:.1e}")
