"""Scratch micro-benchmark: time the gate kernels of several library variants (tools/variants/*.so)."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import _lib, ops, synthetic as syn
from mil_amd.bags import BagLayout

def load(path):
    h = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(h, name); fn.restype = res; fn.argtypes = args
    return h

def timed(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda")
B, N, L = 32, 1024, 512
p = {k: v.to(dev) for k, v in syn.image_only_params(1234, L=L).items()}
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev)
lay = BagLayout.uniform(B, N, dev)
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "variants", "*.so")))
libs = [(os.path.basename(q), load(q)) for q in paths]
w = p["aggregator.attention_weights.weight"].view(-1)
args = (x, p["aggregator.attention_V.0.weight"], p["aggregator.attention_V.0.bias"], p["aggregator.attention_U.0.weight"],
        p["aggregator.attention_U.0.bias"], w, p["aggregator.attention_weights.bias"])
flops = 4.0 * B * N * L * 192
for rnd in range(3):
    for name, h in libs:
        _lib._lib = h
        t_f = timed(lambda: ops.gate_scores_fwd(*args, save_gates=True))
        scores, gates = ops.gate_scores_fwd(*args, save_gates=True)
        if rnd == 0:
            ref_scores = globals().setdefault("REF_SCORES", scores.clone())
            ref_gates = globals().setdefault("REF_GATES", gates.clone())
            print(f"   {name}: max|dscore| vs first lib {float((scores - ref_scores).abs().max()):.2e}  max|dgates| {float((gates - ref_gates).abs().max()):.2e}")
        ds = torch.randn(B * N, device=dev) * 1e-3
        g = [torch.empty_like(p[k]) for k in ("aggregator.attention_V.0.weight", "aggregator.attention_V.0.bias",
             "aggregator.attention_U.0.weight", "aggregator.attention_U.0.bias")] + [torch.empty(192, device=dev), torch.empty(1, device=dev)]
        ws = ops.gate_bwd_params(x, gates, ds, w, *g)
        t_b = timed(lambda: ops.gate_bwd_params(x, gates, ds, w, *g, False, ws))
        print(f"{name:40s} fwd {t_f:8.1f} us ({flops/t_f/1e6:6.1f} TF)   bwd {t_b:8.1f} us ({flops/t_b/1e6:6.1f} TF)", flush=True)
