"""Soak run of the three steppers on a learnable synthetic task (the label is a function of the bag: mean of feature 0 > 0),
thousands of ragged steps each: losses must fall and stay finite, graph counts must stay bounded, parameters finite.
    python tools/soak.py [--steps 2000]"""
import argparse, json, os, sys, time
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.fusion_step import RaggedFusionStepper
from mil_amd.model.utils import get_model
from mil_amd.optim import FlatAdam
from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--lr-scale", type=float, default=1.0, help="multiplies both base learning rates (short runs: 5)")
a = ap.parse_args()
dev = torch.device("cuda")
rng = np.random.default_rng(7)
torch.manual_seed(7)                      # the bags are drawn on the device: same data every run
out = {}


def bag(n, L, cls):
    x = torch.randn((n, L), device=dev)
    x[:, 0] += 0.35 if cls else -0.35            # a weak per-patch signal the pool has to aggregate
    return x


def label(cls):
    return torch.tensor([[1.0, 0.0] if cls == 0 else [0.0, 1.0]], device=dev)


# ---- image-only fused step, ragged buckets, train mode, lr schedule through the graph
tr = ImageOnlyTrainer(syn.image_only_params(1, L=512), dev, lr=3e-4, train_mode=True, counted=True)
st = RaggedImageOnlyStepper(tr, B=1)
losses = []
t0 = time.time()
for i in range(a.steps):
    n, cls = int(rng.integers(2000, 15593)), int(rng.integers(0, 2))
    tr.lr = 3e-4 * a.lr_scale * (0.5 * (1 + np.cos(np.pi * i / a.steps)))
    slot = st.slot(n)
    slot.x[:n].copy_(bag(n, 512, cls))
    slot.y.copy_(label(cls))
    loss, _ = st.step(slot, [n])
    if i % 10 == 0:
        losses.append(float(loss.item()))
torch.cuda.synchronize()
out["image_only"] = {"first": float(np.mean(losses[:10])), "last": float(np.mean(losses[-10:])), "graphs": len(st.slots),
                     "finite": bool(torch.isfinite(tr.fp.flat).all()), "steps_per_s": round(a.steps / (time.time() - t0), 1)}

# ---- fusion model, ragged buckets, train mode, optimizer inside the graph
args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL", num_classes=2,
                       learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9, alignment_base="CI", model_CT="resnetMC3_18",
                       clip_layers=2, cache_text=0)
torch.manual_seed(5)
model = get_model(args).to(dev).train()
opt = FlatAdam([q for q in model.parameters() if q.requires_grad], lr=1e-4, weight_decay=1e-7, counted=True)
fs = RaggedFusionStepper(model, opt, B=1)
ids = syn.make_token_ids(2, 1, 1).to(dev)
losses = []
steps = a.steps // 2
t0 = time.time()
for i in range(steps):
    n, cls = int(rng.integers(2000, 15593)), int(rng.integers(0, 2))
    opt.param_groups[0]["lr"] = 1e-4 * a.lr_scale * (0.5 * (1 + np.cos(np.pi * i / steps)))
    slot = fs.slot(n)
    slot.x[:n].copy_(bag(n, 768, cls))
    slot.y.copy_(label(cls))
    if i == 0:
        fs.encode_notes(slot, ids)
        text = slot.text.clone()
    slot.text.copy_(text)
    loss, _, _ = fs.step(slot, [n])
    if i % 10 == 0:
        losses.append(float(loss))
torch.cuda.synchronize()
out["fusion"] = {"first": float(np.mean(losses[:8])), "last": float(np.mean(losses[-8:])), "graphs": len(fs.gs._graphs),
                 "replays": fs.replays, "finite": bool(torch.isfinite(opt.flat).all()),
                 "adam_steps": int(opt.step_counter.item()), "steps_per_s": round(steps / (time.time() - t0), 1)}
print(json.dumps(out))
ok = all(v["finite"] and v["last"] < v["first"] for v in out.values()) and out["image_only"]["graphs"] <= 8 and out["fusion"]["graphs"] <= 8
sys.exit(0 if ok else 1)
