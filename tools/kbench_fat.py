"""bf16 gate forward at config 5 (32 x 4096 x 1024): the one-wave-per-SIMD kernel (k_gate_fwd_bf16_fat, default) against the
eight-wave deep pipeline (MIL_BF16_FAT=0).  Run once per setting; FAT_DUMP=path saves scores / gates for a bit comparison."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import ops, synthetic as syn

dev = torch.device("cuda")
B, N, L = 32, 4096, 1024
R = B * N
p = {k: v.to(dev) for k, v in syn.image_only_params(1234, L=L).items()}
g = torch.Generator(device="cuda").manual_seed(5)
x16 = ops.cast_bf16(torch.randn((R, L), device=dev, generator=g))
Wv16 = ops.cast_bf16(p["aggregator.attention_V.0.weight"]); Wu16 = ops.cast_bf16(p["aggregator.attention_U.0.weight"])
w = p["aggregator.attention_weights.weight"].view(-1)
args = (x16, Wv16, p["aggregator.attention_V.0.bias"], Wu16, p["aggregator.attention_U.0.bias"], w, p["aggregator.attention_weights.bias"])


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


s, g16 = ops.gate_scores_fwd_bf16(*args, save_gates=True, gates_bf16=True)
torch.cuda.synchronize()
if os.environ.get("FAT_DUMP"):
    torch.save({"s": s.cpu(), "g": g16.cpu()}, os.environ["FAT_DUMP"])
for _ in range(50):
    ops.gate_scores_fwd_bf16(*args, save_gates=True, gates_bf16=True)      # clock ramp
flops = 4.0 * R * L * 192
for save in (True, False):
    t = timed(lambda: ops.gate_scores_fwd_bf16(*args, save_gates=save, gates_bf16=True))
    print(f"MIL_BF16_FAT={os.environ.get('MIL_BF16_FAT', '1')} save_gates={save}: {t:7.1f} us  {flops / t / 1e6:6.0f} TF  finite={bool(torch.isfinite(s).all())}")
