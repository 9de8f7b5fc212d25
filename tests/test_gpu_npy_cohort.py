"""GPU: an on-disk cohort end to end: `.npy` patch-feature bags with the reference's train-time
patch drop and zero-padding (dataset.py:366-393) streamed through train_ddp.py's fused step, against the oracle + torch Adam
on exactly the same bags; then test_ddp.py on the written checkpoint."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mil_amd import synthetic as syn
from mil_amd.dataset import NpyBagDataset, collate_bags
from mil_amd.dist_utils import shard_indices
from oracle import mil_oracle as orc

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "llm-guided-multimodal-mil_amd")


def run(script, *argv):
    r = subprocess.run([sys.executable, os.path.join(PKG, script), *argv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def _cohort(root, L=512):
    rng = np.random.default_rng(7)
    index = {}
    for i, (n, kind) in enumerate([(130, "Biopsy"), (300, "Resection"), (41, "Biopsy"), (222, "Resection"), (64, "Biopsy"),
                                   (97, "Resection")]):
        np.save(os.path.join(root, f"P{i:03d}.npy"), rng.standard_normal((n, L)).astype(np.float32))
        index[f"P{i:03d}"] = {"label": int(i % 3 == 0), "kind": kind, "ids": [[49406, 11, 12, 49407] + [0] * 73]}
    with open(os.path.join(root, "index.json"), "w") as f:
        json.dump(index, f)
    return index


@pytest.mark.parametrize("graph", [0, 1])
def test_three_steps_from_npy_bags_match_oracle_adam(tmp_path, graph):
    root = str(tmp_path / "cohort")
    os.makedirs(root)
    index = _cohort(root)
    out = str(tmp_path / "ck")
    seed, steps, bs = 1234, 3, 2
    run("train_ddp.py", "--variant", "image_only", "--fused_step", "--no_dropout", "--path_data_pathology", root,
        "--batch_size", str(bs), "--n_epochs", "1", "--iter_per_epoch", str(steps), "--seed", str(seed), "--hip_graph", str(graph),
        "--save_dir", out)
    ck = torch.load(os.path.join(out, "checkpoint_best.pth.tar"), weights_only=True)
    # the same stream on the CPU: same dataset class (seeded patch drop), same sampler, oracle + torch.optim.Adam
    ds = NpyBagDataset(root, index, mode="train", augmentation=True, num_classes=2, seed=seed)
    order = shard_indices(len(ds), 1, 0, epoch=0)
    torch.manual_seed(seed)
    from types import SimpleNamespace
    from mil_amd.model.utils_clip import get_model
    m = get_model(SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", num_classes=2, patch_dim=512))
    p = {k.replace("extractor_pathology.", "aggregator."): v.detach().clone() for k, v in m.state_dict().items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-7)
    for it in range(steps):
        batch = collate_bags([ds[j] for j in order[it * bs:(it + 1) * bs]])
        bags = [batch["pathology"][b, :n] for b, n in enumerate(batch["lengths"])]
        prob = torch.cat([orc.image_only_forward(xb, ref)["prob"] for xb in bags], 0)
        loss = orc.bce_loss(prob, batch["label"])
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert ck["optimizer"]["step"] == steps
    for k, v in ref.items():
        km = k.replace("aggregator.", "extractor_pathology.")
        if k.endswith("attention_weights.bias"):
            continue                                        # softmax bias: exactly-zero gradient, Adam amplifies rounding noise
        assert float((ck["state_dict"][km].cpu() - v.detach()).abs().max()) <= 2e-7, k
        assert float((ck["state_dict"][km].cpu() - p[k]).abs().max()) > 1e-6, k      # it did train
    out_t = run("test_ddp.py", "--variant", "image_only", "--path_data_pathology", root, "--test_pth", out)
    assert "bags 6" in out_t and "Time for inference" in out_t
