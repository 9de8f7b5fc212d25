"""Synthetic cohort for the entry points.  The reference's dataset.py (Excel cohorts, DICOM/NIfTI CT, `.npy`
pathology bags, CLIP-tokenised notes) needs private hospital data and is out of scope; this generator yields
items of the same shape: a bag of patch features [n, F] fp32, a tokenised note int64 [P, 77] and a one-hot label."""
from typing import List

import torch
from torch.utils.data import Dataset

from . import synthetic as syn


class SyntheticBags(Dataset):
    def __init__(self, n_bags: int, patches: int, feat_dim: int, prompts: int = 1, num_classes: int = 2,
                 seed: int = 1234, ragged: bool = False, keep: float = 1.0):
        self.keep = float(keep)              # per-epoch patch drop of a resident cohort (cohort.DeviceCohort.from_dataset)
        self.n, self.patches, self.feat_dim, self.prompts, self.C, self.seed = n_bags, patches, feat_dim, prompts, num_classes, seed
        g = torch.Generator().manual_seed(seed)
        lo = max(1, patches // 2)
        self.lengths: List[int] = ([int(v) for v in torch.randint(lo, patches + 1, (n_bags,), generator=g)]
                                   if ragged else [patches] * n_bags)
        self.labels = syn.make_labels(seed + 1, n_bags, num_classes)
        self.ids = syn.make_token_ids(seed + 2, n_bags, prompts)

    def __len__(self):
        return self.n

    def __getitem__(self, i: int):
        x = torch.randn((self.lengths[i], self.feat_dim), generator=torch.Generator().manual_seed(self.seed + 1000 + i))
        return {"pathology": x, "CI": self.ids[i], "label": self.labels[i], "length": self.lengths[i], "index": i}


def load_cohort(args, mode: str, prompts: int):
    """The dataset the entry points iterate: on-disk `.npy` bags when --path_data_pathology is given (dataset.py:366-393),
    synthetic bags otherwise.  Returns (dataset, patch feature width)."""
    import json
    import os
    root = getattr(args, "path_data_pathology", "")
    if root:
        idx_path = getattr(args, "index_json", "") or os.path.join(root, "index.json")
        with open(idx_path) as f:
            index = json.load(f)
        ds = NpyBagDataset(root, index, mode=mode, augmentation=bool(getattr(args, "augmentation", 1)),
                           num_classes=args.num_classes, seed=args.seed)
        import numpy as np
        first = np.load(os.path.join(root, ds.keys[0] + ".npy"), mmap_mode="r")       # header only: feature width
        return ds, int(first.shape[1])
    n_patch, feat, n_bags = [int(v) for v in args.synthetic]
    seed = args.seed + (0 if mode == "train" else 1)
    return SyntheticBags(n_bags, n_patch, feat, prompts, args.num_classes, seed, args.ragged,
                         keep=float(getattr(args, "patch_keep", 1.0)) if mode == "train" else 1.0), feat


def collate_bags(items):
    """Zero-pad to the longest bag of the batch (dataset.py:386-391 pads to a fixed length when batch > 1) and
    return the true lengths alongside."""
    n_max = max(it["length"] for it in items)
    x = torch.zeros((len(items), n_max, items[0]["pathology"].shape[1]))
    for b, it in enumerate(items):
        x[b, :it["length"]] = it["pathology"][:it["length"]]
    return {"pathology": x, "CI": torch.stack([it["CI"] for it in items]),
            "label": torch.stack([it["label"] for it in items]), "lengths": [it["length"] for it in items]}


class NpyBagDataset(Dataset):
    """On-disk pathology bags as the reference stores them: one `<patientid>.npy` per patient holding the
    CTransPath features [n, 768] fp32 (dataset.py:366-367).  Mirrors the train-time patch-drop augmentation
    (a sorted random subset of 90 % of the patches for biopsies, 80 % for resections, dataset.py:374-381) and the
    zero-padding to a fixed row count used when batch_size > 1 (:383-391).  Labels/notes come from a small JSON
    index {"id": {"label": 0|1, "kind": "Biopsy"|"Resection", "ids": [[...77 ints...], ...]}} instead of the
    private Excel sheets."""

    def __init__(self, root: str, index: dict, mode: str = "train", augmentation: bool = True, pad_to: int = 0,
                 num_classes: int = 2, seed: int = 1234):
        import random
        self.root, self.mode, self.aug, self.pad_to, self.C = root, mode, augmentation, pad_to, num_classes
        self.keys = sorted(index.keys())
        self.index = index
        self.rng = random.Random(seed)

    def __len__(self):
        return len(self.keys)

    def __getitem__(self, i: int):
        import os
        import numpy as np
        key = self.keys[i]
        meta = self.index[key]
        feat = np.load(os.path.join(self.root, key + ".npy"))               # allow_pickle stays False
        n = feat.shape[0]
        if self.mode == "train" and self.aug:
            keep = 0.9 if meta.get("kind", "Biopsy") == "Biopsy" else 0.8
            sel = sorted(self.rng.sample(range(n), int(n * keep)))
            feat = feat[sel, :]
        length = feat.shape[0]
        if self.pad_to:
            out = np.zeros((self.pad_to, feat.shape[1]), dtype=np.float32)
            out[:length] = feat
            feat = out
        label = torch.nn.functional.one_hot(torch.tensor(int(meta["label"])), self.C).float()
        ids = torch.tensor(meta.get("ids", [[0] * 77]), dtype=torch.int64)
        return {"pathology": torch.from_numpy(np.ascontiguousarray(feat)).float(), "CI": ids, "label": label,
                "length": length, "index": i}
