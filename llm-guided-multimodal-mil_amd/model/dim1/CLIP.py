"""Clinical-text extractor (reference: model/dim1/CLIP.py:7-77).

`clip.load("ViT-B/32")` downloads weights (clip/clip.py:34,55) and cannot run offline; the tower is built with
the ViT-B/32 text architecture and either seeded random weights or a state_dict the caller loads.

Two branches, as upstream:
  * frozen (`learnablePrompt == 0`, :71-75): `encode_text` under no_grad, optionally cached per note;
  * learnable context / CoOp (`learnablePrompt == 1`, :29-62): `ctx [len(clinical_features)+1, n_ctx, W]` replaces
    token positions 1 .. n_ctx of every prompt and is trained THROUGH the frozen tower (the HIP kernels provide the
    tower's backward: GEMM dx, LayerNorm, causal attention, QuickGELU).  Upstream reads only `x[0]` and returns
    `[P, 512]` (it assumes one bag per forward); here every bag of the batch gets its own `[P, 512]` -> `[B, P, 512]`."""
import torch
import torch.nn as nn

from ...clip.model import CLIPText


class CLIP(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.model = CLIPText(int(getattr(args, "clip_embed", 512)), 77, int(getattr(args, "clip_vocab", 49408)), int(getattr(args, "clip_width", 512)),
                              int(getattr(args, "clip_heads", 8)), int(getattr(args, "clip_layers", 12)))
        for p in self.model.parameters():
            p.requires_grad_(False)
        # opt-in: split-bf16 products for the frozen tower's GEMMs (clip/model.py: set_gemm_pieces); default fp32
        self.model.set_gemm_pieces(int(getattr(args, "clip_gemm_pieces", 0)))
        self._cache = {}
        self.cache_limit = int(getattr(args, "cache_text_limit", 65536))     # notes kept (2 KB each)
        if getattr(args, "learnablePrompt", 0):
            self.ctx_dim = self.model.ln_final.weight.shape[0]
            n_prompts = len(getattr(args, "clinical_features", [])) + 1                   # CLIP.py:19
            ctx = torch.empty(n_prompts, int(getattr(args, "n_ctx", 8)), self.ctx_dim)
            nn.init.normal_(ctx, std=0.02)                                                # CLIP.py:21
            self.ctx = nn.Parameter(ctx)

    def _forward_learnable(self, x: torch.Tensor) -> torch.Tensor:
        B, P, ctx_len = x.shape
        n_ctx = self.ctx.shape[1]
        if P != self.ctx.shape[0]:
            raise ValueError(f"learnable prompts: got {P} prompts per bag, ctx was built for {self.ctx.shape[0]} "
                             "(len(clinical_features) + 1)")
        with torch.no_grad():
            emb = self.model.token_embedding(x)                                           # [B, P, 77, W]  (:33)
        prompts = torch.cat([emb[:, :, :1], self.ctx.unsqueeze(0).expand(B, -1, -1, -1),
                             emb[:, :, 1 + n_ctx:]], dim=2)                               # (:45-52)
        feats = self.model.encode_embedded(prompts.reshape(B * P, ctx_len, -1), x.reshape(B * P, ctx_len))
        return feats.reshape(B, P, -1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x int64 [B, P, ctx] tokenised notes -> [B, P, 512]  (dim1/CLIP.py:71-77, one encode_text over all B*P
        prompts instead of a Python loop over the batch)."""
        if getattr(self.args, "learnablePrompt", 0):
            return self._forward_learnable(x)
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
        first = {}
        for i, k in enumerate(keys):
            if k not in self._cache:
                first.setdefault(k, i)
        if first:
            idx = torch.tensor(list(first.values()), device=x.device)
            with torch.no_grad():
                new = self.model.encode_text(flat.index_select(0, idx))
            if len(self._cache) + len(first) > self.cache_limit:      # bounded: a cohort larger than the limit re-encodes
                self._cache = {k: v for k, v in self._cache.items() if k in set(keys)}
            for j, k in enumerate(first.keys()):
                self._cache[k] = new[j].clone()
        return torch.stack([self._cache[k] for k in keys], 0).reshape(B, P, -1)
