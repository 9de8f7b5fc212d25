"""Time the bf16 gate kernels alone at config 5 (32 bags x 4096 patches x 1024), HIP events around 20 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops, synthetic as syn

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
p = {k: v.to(dev) for k, v in syn.image_only_params(1234, L=L).items()}
x16 = ops.cast_bf16(torch.randn((R, L), device=dev))
Wv16 = ops.cast_bf16(p["aggregator.attention_V.0.weight"]); Wu16 = ops.cast_bf16(p["aggregator.attention_U.0.weight"])
w = p["aggregator.attention_weights.weight"].view(-1)
args = (x16, Wv16, p["aggregator.attention_V.0.bias"], Wu16, p["aggregator.attention_U.0.bias"], w, p["aggregator.attention_weights.bias"])
flops = 4.0 * R * L * 192
import glob, ctypes
from mil_amd import _lib
def load(path):
    h = ctypes.CDLL(path)
    for name, (res, a) in _lib.SIGNATURES.items():
        fn = getattr(h, name); fn.restype = res; fn.argtypes = a
    return h
for q in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so"))):
    keep = _lib._lib
    _lib._lib = load(q)
    t = timed(lambda: ops.gate_scores_fwd_bf16(*args, save_gates=False))
    sv, gv = ops.gate_scores_fwd_bf16(*args, save_gates=True)
    _lib._lib = keep
    sm, gm = ops.gate_scores_fwd_bf16(*args, save_gates=True)
    print(f"variant {os.path.basename(q):28s} fwd (no gates) {t:7.1f} us   max|dscore| vs main {float((sv - sm).abs().max()):.1e} max|dgates| {float((gv - gm).abs().max()):.1e}")
for save in (True, False):
    t = timed(lambda: ops.gate_scores_fwd_bf16(*args, save_gates=save))
    print(f"gate_fwd_bf16 R={R} L={L} save_gates={save}: {t:7.1f} us  {flops / t / 1e6:7.1f} TF  x-stream {R * L * 2 / t / 1e6:5.2f} TB/s")
# consistency: the 256-row deep kernel (R >= 65536) against the 128-row kernel on a slice
s_all, g_all = ops.gate_scores_fwd_bf16(*args, save_gates=True)
s_sl, g_sl = ops.gate_scores_fwd_bf16(x16[:4096].contiguous(), *args[1:], save_gates=True)
print("max|dscore| deep vs 128-row:", float((s_all[:4096] - s_sl).abs().max()), " max|dgates|:", float((g_all[:4096] - g_sl).abs().max()))
