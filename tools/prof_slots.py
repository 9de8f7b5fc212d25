"""Which parameter gradients of the fusion step are NOT written straight into optim.FlatAdam's flat buffer (each costs a copy
in FlatAdam.gather - a memcpy node per parameter inside a captured step)?"""
import os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.model.utils import get_model
from mil_amd.optim import FlatAdam

dev = torch.device("cuda")
args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
                       learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9, alignment_base="CI", model_CT="resnetMC3_18",
                       clip_layers=1, cache_text=1)
torch.manual_seed(1234)
model = get_model(args).to(dev).eval()
B = int(os.environ.get("BAGS", "32"))
x = syn.make_bags(1, B, 1024, 768).to(dev)
ids = syn.make_token_ids(2, B, 1).to(dev)
y = syn.make_labels(3, B).to(dev)
opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-5, weight_decay=1e-7, counted=True)
names = {id(p): k for k, p in model.named_parameters()}
with torch.no_grad():
    tfeat = model.clinic_extractor(ids)
for _ in range(2):
    opt.zero_grad()
    model([x], ids, labels=y, text_features=tfeat)
    model.last_loss.backward()
torch.cuda.synchronize()
n = 0
for p, slot in zip(opt.params, opt._gviews):
    if p.grad is not None and p.grad.data_ptr() != slot.data_ptr():
        n += 1
        print("copied:", names[id(p)], tuple(p.shape))
print(n, "of", sum(p.grad is not None for p in opt.params), "live gradients need a gather copy")
