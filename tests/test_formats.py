"""CPU: on-disk formats either side of the path: `.npy` patch-feature bags with the
reference's drop/pad behaviour, and checkpoint round trip with the reference's state_dict key schema."""
from types import SimpleNamespace

import numpy as np
import torch

from mil_amd.dataset import NpyBagDataset, collate_bags
from mil_amd.model.utils import get_model
from mil_amd.utils import save_checkpoint


def test_npy_bag_dataset_drop_and_pad(tmp_path):
    rng = np.random.default_rng(0)
    index = {}
    for i, (n, kind) in enumerate([(50, "Biopsy"), (200, "Resection"), (7, "Biopsy")]):
        np.save(tmp_path / f"P{i:03d}.npy", rng.standard_normal((n, 768)).astype(np.float32))
        index[f"P{i:03d}"] = {"label": i % 2, "kind": kind, "ids": [[49406, 5, 6, 49407] + [0] * 73]}
    ev = NpyBagDataset(str(tmp_path), index, mode="test")
    assert [ev[i]["length"] for i in range(3)] == [50, 200, 7]
    tr = NpyBagDataset(str(tmp_path), index, mode="train", augmentation=True, pad_to=256)
    it = [tr[i] for i in range(3)]
    assert [d["length"] for d in it] == [45, 160, 6]                  # int(n * 0.9), int(n * 0.8)
    full1 = np.load(tmp_path / "P001.npy")
    kept = it[1]["pathology"][:160].numpy()
    # the kept rows are a sorted subset of the original rows
    pos = [int(np.where((full1 == r).all(1))[0][0]) for r in kept[:10]]
    assert pos == sorted(pos)
    assert tuple(it[1]["pathology"].shape) == (256, 768) and float(it[1]["pathology"][160:].abs().sum()) == 0.0
    b = collate_bags(it)
    assert b["lengths"] == [45, 160, 6] and tuple(b["CI"].shape) == (3, 1, 77)


def test_checkpoint_roundtrip_keeps_reference_schema(tmp_path):
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, clip_layers=1, clip_vocab=512)
    torch.manual_seed(0)
    m = get_model(args)
    save_checkpoint({"epoch": 3, "state_dict": m.state_dict()}, True, str(tmp_path), "checkpoint_0002.pth.tar")
    ck = torch.load(tmp_path / "checkpoint_best.pth.tar", weights_only=True)        # test_ddp.py:89-99
    torch.manual_seed(1)
    m2 = get_model(args)
    missing, unexpected = m2.load_state_dict(ck["state_dict"], strict=True)
    assert not missing and not unexpected and ck["epoch"] == 3
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
