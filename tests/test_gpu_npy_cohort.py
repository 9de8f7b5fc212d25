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
        "--resident_cohort", "0", "--save_dir", out)        # the host pipeline: np.load + random.sample drop per step
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


def test_resident_cohort_run_matches_oracle_adam_on_the_rows_the_device_drew(tmp_path):
    """`train_ddp.py --hip_graph 1` on an on-disk cohort, default input side: the bags are loaded once into HBM, the epoch's
    patch drop is drawn on the device (Philox) and every step is fed by one gather launch (cohort.DeviceCohort).  Two
    epochs x three steps against the oracle + torch.optim.Adam on the rows oracle/cohort.py says were drawn."""
    from oracle import cohort as oc
    root = str(tmp_path / "cohort")
    os.makedirs(root)
    index = _cohort(root)
    out = str(tmp_path / "ck")
    seed, steps, bs, epochs = 1234, 3, 2, 2
    log = run("train_ddp.py", "--variant", "image_only", "--fused_step", "--no_dropout", "--path_data_pathology", root,
              "--batch_size", str(bs), "--n_epochs", str(epochs), "--iter_per_epoch", str(steps), "--seed", str(seed),
              "--hip_graph", "1", "--save_dir", out)
    assert "cohort resident in HBM: 6 bags" in log
    ck = torch.load(os.path.join(out, "checkpoint_best.pth.tar"), weights_only=True)
    keys = sorted(index)
    bags = [torch.from_numpy(np.load(os.path.join(root, k + ".npy"))) for k in keys]
    keep = [0.9 if index[k]["kind"] == "Biopsy" else 0.8 for k in keys]
    labels = torch.nn.functional.one_hot(torch.tensor([index[k]["label"] for k in keys]), 2).float()
    torch.manual_seed(seed)
    from types import SimpleNamespace
    from mil_amd.model.utils_clip import get_model
    m = get_model(SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", num_classes=2, patch_dim=512))
    p = {k.replace("extractor_pathology.", "aggregator."): v.detach().clone() for k, v in m.state_dict().items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-7)
    for epoch in range(epochs):
        order = shard_indices(len(keys), 1, 0, epoch=epoch)
        for it in range(steps):
            take = order[it * bs:(it + 1) * bs]
            xs = [bags[j][oc.patch_drop_select(bags[j].shape[0], int(bags[j].shape[0] * keep[j]), j, seed, epoch)] for j in take]
            prob = torch.cat([orc.image_only_forward(xb, ref)["prob"] for xb in xs], 0)
            loss = orc.bce_loss(prob, labels[take])
            opt.zero_grad()
            loss.backward()
            opt.step()
    assert ck["optimizer"]["step"] == steps * epochs
    for k, v in ref.items():
        km = k.replace("aggregator.", "extractor_pathology.")
        if k.endswith("attention_weights.bias"):
            continue
        assert float((ck["state_dict"][km].cpu() - v.detach()).abs().max()) <= 4e-7, k
        assert float((ck["state_dict"][km].cpu() - p[k]).abs().max()) > 1e-6, k
