"""Synthetic cohort for the entry points.  The reference's dataset.py (Excel cohorts, DICOM/NIfTI CT, `.npy`
pathology bags, CLIP-tokenised notes) needs private hospital data and is out of scope; this generator yields
items of the same shape: a bag of patch features [n, F] fp32, a tokenised note int64 [P, 77] and a one-hot label."""
from typing import List

import torch
from torch.utils.data import Dataset

from . import synthetic as syn


class SyntheticBags(Dataset):
    def __init__(self, n_bags: int, patches: int, feat_dim: int, prompts: int = 1, num_classes: int = 2,
                 seed: int = 1234, ragged: bool = False):
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


def collate_bags(items):
    """Zero-pad to the longest bag of the batch (dataset.py:386-391 pads to a fixed length when batch > 1) and
    return the true lengths alongside."""
    n_max = max(it["length"] for it in items)
    x = torch.zeros((len(items), n_max, items[0]["pathology"].shape[1]))
    for b, it in enumerate(items):
        x[b, :it["length"]] = it["pathology"]
    return {"pathology": x, "CI": torch.stack([it["CI"] for it in items]),
            "label": torch.stack([it["label"] for it in items]), "lengths": [it["length"] for it in items]}
