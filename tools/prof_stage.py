"""Scratch: run the image-only step a few times (for rocprofv3 --pmc / --kernel-trace of single stages)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer
dev = torch.device("cuda")
bf16 = "--bf16" in sys.argv
B = 32
N, L = (4096, 1024) if bf16 else (1024, 512)
tr = ImageOnlyTrainer(syn.image_only_params(1234, L=L), dev, train_mode=("--eval" not in sys.argv) and not bf16)
x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev)
if bf16:
    x = x.to(torch.bfloat16)
y = syn.make_labels(99, B).to(dev)
lay = BagLayout.uniform(B, N, dev)
for _ in range(int(os.environ.get("STEPS", "12"))):
    tr.train_step(x, lay, y)
torch.cuda.synchronize()
print("done")
