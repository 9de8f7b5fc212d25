"""Scratch: loop / epilogue cycles and in-kernel clock of k_gate_fwd_bf16_deep (needs the -DHC_STAMP variant)."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import _lib, synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
q = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*HC_STAMP*.so"))[0]
h = ctypes.CDLL(q)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(h, name); fn.restype = res; fn.argtypes = args
_lib._lib = h
dev = torch.device("cuda")
B, N, L = 32, 4096, 1024
tr = ImageOnlyTrainer(syn.image_only_params(1234, L=L), dev)
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev).to(torch.bfloat16)
y = syn.make_labels(99, B).to(dev)
lay = BagLayout.uniform(B, N, dev)
for steps in (20, 200):
    for _ in range(steps):
        tr.forward(x, lay, y)
    torch.cuda.synchronize()
    g = tr.last["gates"].view(-1).view(torch.float32)[:4 * 512].view(512, 4).cpu()
    clk = g[:, 0] / g[:, 1] * 100
    t0 = g[:, 3]
    first, second = g[t0 <= t0.median()], g[t0 > t0.median()]
    print(f"after {steps}: loop cycles median {float(g[:,0].median()):.0f} (ideal 2 waves x 768 MFMA x 32 = 49152), epilogue cycles {float(g[:,2].median()):.0f}, "
          f"clock {float(clk.median()):.0f} MHz, loop us {float((g[:,1]/100).median()):.1f}; start-tick spread {float(t0.max()-t0.min())/100:.1f} us")
