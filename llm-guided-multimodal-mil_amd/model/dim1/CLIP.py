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
        with torch.no_grad():
            feats = self.model.encode_text(x.reshape(B * P, ctx))
        return feats.reshape(B, P, -1)
