"""Clinical-text extractor (reference: model/dim1/CLIP.py:7-77), frozen branch.

`clip.load("ViT-B/32")` downloads weights (clip/clip.py:34,55) and cannot run offline; the tower is built with
the ViT-B/32 text architecture and either seeded random weights or a state_dict the caller loads.  The
learnable-prompt (CoOp) branch (:29-62) needs a backward through the tower and is listed as "next" (SURVEY 8f)."""
import torch
import torch.nn as nn

from ...clip.model import CLIPText


class CLIP(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if getattr(args, "learnablePrompt", 0):
            raise NotImplementedError("learnable prompts (CoOp) need the text tower's backward: not on the built path yet")
        self.model = CLIPText(512, 77, int(getattr(args, "clip_vocab", 49408)), int(getattr(args, "clip_width", 512)),
                              int(getattr(args, "clip_heads", 8)), int(getattr(args, "clip_layers", 12)))
        for p in self.model.parameters():
            p.requires_grad_(False)
        self._cache = {}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x int64 [B, P, ctx] tokenised notes -> [B, P, 512]  (dim1/CLIP.py:71-77, one encode_text over all B*P
        prompts instead of a Python loop over the batch)."""
        B, P, ctx = x.shape
        flat = x.reshape(B * P, ctx)
        if not getattr(self.args, "cache_text", 0):
            with torch.no_grad():
                feats = self.model.encode_text(flat)
            return feats.reshape(B, P, -1)
        # The tower is frozen, so a note's embedding never changes: encode each distinct token row once
        # (keyed by its bytes; costs one small device->host copy of the ids per call) and replay it afterwards.
        host = flat.cpu().numpy()
        keys = [row.tobytes() for row in host]
        todo = sorted({i for i, k in enumerate(keys) if k not in self._cache}, key=lambda i: i)
        first = {}
        for i in todo:
            first.setdefault(keys[i], i)
        if first:
            idx = torch.tensor(list(first.values()), device=x.device)
            with torch.no_grad():
                new = self.model.encode_text(flat.index_select(0, idx))
            for j, k in enumerate(first.keys()):
                self._cache[k] = new[j].clone()
        return torch.stack([self._cache[k] for k in keys], 0).reshape(B, P, -1)
