"""Scratch driver for rocprofv3: the bucketed fusion step on ONE bucket (default 12 288 rows, a bag of 10 000 patches), 40
replays - per-kernel averages of the authors' regime (one ragged bag per GPU)."""
import os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.fusion_step import RaggedFusionStepper
from mil_amd.model.utils import get_model
from mil_amd.optim import FlatAdam

dev = torch.device("cuda")
n = int(os.environ.get("PATCHES", "10000"))
args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
                       learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9, alignment_base="CI", model_CT="resnetMC3_18",
                       clip_layers=1, cache_text=0)
torch.manual_seed(1234)
model = get_model(args).to(dev).train()
opt = FlatAdam([q for q in model.parameters() if q.requires_grad], lr=1e-5, weight_decay=1e-7, counted=True)
st = RaggedFusionStepper(model, opt, B=1)
x = torch.randn((n, 768), device=dev)
slot = st.slot(n)
slot.x[:n].copy_(x)
slot.y.copy_(syn.make_labels(3, 1).to(dev))
st.encode_notes(slot, syn.make_token_ids(2, 1, 1).to(dev))
for _ in range(int(os.environ.get("STEPS", "40"))):
    st.step(slot, [n])
torch.cuda.synchronize()
print("done", st.replays)
