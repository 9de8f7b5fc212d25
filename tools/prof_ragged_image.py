"""Scratch driver for rocprofv3: the bucketed image-only step on ONE bucket (a bag of PATCHES rows), STEPS replays."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper
dev = torch.device("cuda")
n = int(os.environ.get("PATCHES", "10000"))
tr = ImageOnlyTrainer(syn.image_only_params(1, L=512), dev, train_mode=True, counted=True)
st = RaggedImageOnlyStepper(tr, B=1)
slot = st.slot(n)
slot.x[:n].copy_(torch.randn((n, 512), device=dev))
slot.y.copy_(syn.make_labels(3, 1).to(dev))
for _ in range(int(os.environ.get("STEPS", "60"))):
    st.step(slot, [n])
torch.cuda.synchronize()
print("done", st.replays)
