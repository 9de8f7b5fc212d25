"""Scratch: in-kernel clock of the dW main loop (needs the -DGB_STAMP variant): s_memtime / s_memrealtime * 100 MHz."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import _lib, synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
q = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*GB_STAMP*.so"))[0]
h = ctypes.CDLL(q)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(h, name); fn.restype = res; fn.argtypes = args
_lib._lib = h
dev = torch.device("cuda")
B, N, L = 32, 1024, 512
tr = ImageOnlyTrainer(syn.image_only_params(1234, L=L), dev, train_mode="--eval" not in sys.argv)
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev)
y = syn.make_labels(99, B).to(dev)
lay = BagLayout.uniform(B, N, dev)
for steps in (1, 50, 400, 2000):
    for _ in range(steps):
        tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    ws = tr._ws["dw_ws"]
    S, NJ = 21, 4
    pb = ws[S * 384 * L:].view(-1, 4, 192)[:S, 3, 8:8 + 24].reshape(S, 12, 2).cpu()
    bar = ws[S * 384 * L:].view(-1, 4, 192)[:S, 3, 64:72].cpu()
    print("   barrier cycles per wave (median over chunks):", [int(v) for v in bar.median(0).values])
    clk = (pb[..., 0] / pb[..., 1] * 100.0)
    print(f"after {steps:5d} more steps: loop cycles median {float(pb[..., 0].median()):.0f}  clock MHz median {float(clk.median()):.0f} min {float(clk.min()):.0f} max {float(clk.max()):.0f}  loop us {float((pb[..., 1] / 100.0).median()):.1f}")
