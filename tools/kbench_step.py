"""Scratch: A/B the image-only step's stages across library builds (main + tools/variants/*.so), interleaved rounds in one
process, HIP-event timing from C (mil_image_only_step_time).  Usage: kbench_step.py [--bf16] [--eval] [--rounds 3]"""
import argparse, ctypes, glob, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import _lib, synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--eval", action="store_true")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--bags", type=int, default=32)
ap.add_argument("--patches", type=int, default=0)
ap.add_argument("--dim", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
B = a.bags
N = a.patches or (4096 if a.bf16 else 1024)
L = a.dim or (1024 if a.bf16 else 512)


def load(path):
    h = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
    return h


main = _lib.lib()
libs = [("main", main)] + [(os.path.basename(q)[3:-3], load(q)) for q in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so")))]
p = syn.image_only_params(1234, L=L)
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev)
if a.bf16:
    x = x.to(torch.bfloat16)
y = syn.make_labels(99, B).to(dev)
lay = BagLayout.uniform(B, N, dev)
res, grads = {}, {}
for rnd in range(a.rounds):
    for name, h in libs:
        _lib._lib = h
        tr = ImageOnlyTrainer(p, dev, train_mode=not a.eval and not a.bf16)
        kb = tr.time_pieces(x, lay, y, iters=30, warm=5)
        tr.reset_dropout_stream()
        tr.forward(x, lay, y)
        tr.backward()
        torch.cuda.synchronize()
        grads.setdefault(name, tr.fp.grad.clone())
        for k, v in kb.items():
            res.setdefault((name, k), []).append(v * 1e3)
_lib._lib = main
keys = [k for (n, k) in res if n == "main"]
flops = 4.0 * B * N * L * 192
print(f"{'lib':28s} " + " ".join(f"{k[:14]:>14s}" for k in keys) + "   max|dgrad| vs main")
for name, _ in libs:
    row = " ".join(f"{statistics.median(res[(name, k)]):8.1f}/{min(res[(name, k)]):5.1f}" for k in keys)
    d = float((grads[name] - grads["main"]).abs().max())
    print(f"{name:28s} {row}   {d:.2e}")
for name, _ in libs:
    f, b = statistics.median(res[(name, 'gate_fwd')]), statistics.median(res[(name, 'gate_bwd_dw')])
    print(f"{name:28s} fwd {flops / f / 1e6:7.1f} TF   dw {flops / b / 1e6:7.1f} TF")
