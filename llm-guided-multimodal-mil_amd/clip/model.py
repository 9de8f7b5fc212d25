"""CLIP text tower (reference: clip/model.py, text side only: encode_text :339-352, Transformer /
ResidualAttentionBlock :167-199, LayerNorm :153-159, QuickGELU :162-164, causal mask :324-330).

Forward-only (the tower is frozen: model/dim1/CLIP.py:71-75 runs it under no_grad).  Parameter names match
the reference state_dict (token_embedding.weight, positional_embedding, transformer.resblocks.N.{attn.in_proj_*,
attn.out_proj.*, ln_1.*, mlp.c_fc.*, mlp.c_proj.*, ln_2.*}, ln_final.*, text_projection).  The visual tower is
never run on this path and is not instantiated (its keys are ignored when loading a full CLIP state_dict)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import lifetime, ops
from ..segments import AttnSegs


class _Attn(nn.Module):          # parameter container with nn.MultiheadAttention's names
    def __init__(self, width: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = _Attn(d_model)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)
        self.n_head = n_head
        self.gemm_pieces = 0            # 0: fp32 MFMA GEMM (default); 2 / 3: split-bf16 products for the FROZEN weights
        self._split = {}

    def _fs(self, name: str, W):
        """bf16 pieces of a frozen weight, formed once (re-formed if the tensor is replaced or modified)."""
        hit = self._split.get(name)
        if hit is None or hit.pieces != self.gemm_pieces or hit.key != (W.data_ptr(), W._version):
            hit = self._split[name] = ops.FrozenSplit(W, self.gemm_pieces)
        return hit

    def _split_ok(self, x) -> bool:
        frozen = not (self.attn.in_proj_weight.requires_grad or self.mlp.c_fc.weight.requires_grad)
        return self.gemm_pieces in (2, 3) and frozen and x.shape[0] > ops.SMALL_ROWS

    def flat(self, x, segs):
        W = x.shape[1]
        # layer_norm_res hands x back as `xr`: the residual adds read xr, so the gradient of the skip branch is added
        # inside the LayerNorm backward kernel
        h, xr = ops.layer_norm_res(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        if self._split_ok(x) and ops.seq_attention_ok(segs):
            # opt-in: the four frozen-weight products (and their dx halves) on the split-bf16 GEMM (csrc/linear_x.hip)
            qkv = ops.linear_frozen_split(h, self._fs("in_proj", w), b)
            o = ops.attention_seq_packed(qkv, segs, self.n_head, causal=True)
            x = ops.linear_frozen_split(o, self._fs("out_proj", self.attn.out_proj.weight), self.attn.out_proj.bias, residual=xr)
            h, xr = ops.layer_norm_res(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
            return ops.mlp_quickgelu_frozen_split(h, self._fs("c_fc", self.mlp.c_fc.weight), self.mlp.c_fc.bias,
                                                  self._fs("c_proj", self.mlp.c_proj.weight), self.mlp.c_proj.bias, residual=xr)
        if ops.seq_attention_ok(segs):
            # one in_proj GEMM of width 3 W; the attention kernels read its column blocks in place
            o = ops.attention_seq_packed(ops.linear_act(h, w, b), segs, self.n_head, causal=True)
        else:
            q = ops.linear_act(h, w[:W], b[:W])
            k = ops.linear_act(h, w[W:2 * W], b[W:2 * W])
            v = ops.linear_act(h, w[2 * W:], b[2 * W:])
            o = ops.attention_rows(q, k, v, segs, self.n_head, causal=True)
        x = ops.linear_act(o, self.attn.out_proj.weight, self.attn.out_proj.bias, "none", residual=xr)
        h, xr = ops.layer_norm_res(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        return ops.mlp_quickgelu(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias,
                                 residual=xr)


def _flat_last_rows(self, x, last, segs):
    """This block's output at the rows `last` (one per sequence, the final row of its prefix) - clip/model.py:183-199
    restricted to the rows that are read out."""
    W = x.shape[1]
    h = ops.layer_norm(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
    w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
    k = ops.linear_act(h, w[W:2 * W], b[W:2 * W])
    v = ops.linear_act(h, w[2 * W:], b[2 * W:])
    q = ops.linear_act(h.index_select(0, last), w[:W], b[:W])
    o = ops.attention_pool(q, k, v, segs, self.n_head)
    x_e = ops.linear_act(o, self.attn.out_proj.weight, self.attn.out_proj.bias, "none", residual=x.index_select(0, last))
    h_e = ops.layer_norm(x_e, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
    return ops.mlp_quickgelu(h_e, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias,
                             residual=x_e)


ResidualAttentionBlock.flat_last_rows = _flat_last_rows


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class CLIPText(nn.Module):
    def __init__(self, embed_dim: int = 512, context_length: int = 77, vocab_size: int = 49408,
                 transformer_width: int = 512, transformer_heads: int = 8, transformer_layers: int = 12):
        super().__init__()
        self.context_length = context_length
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600369327779)      # ln(1/0.07), clip/model.py:291
        self._proj_t = None
        self.static_rows = False        # True: fixed-shape tower (graph-capturable), see _tower_static
        self.initialize_parameters()

    def initialize_parameters(self):
        w, n = self.transformer.width, self.transformer.layers
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=w ** -0.5)
            nn.init.normal_(blk.attn.out_proj.weight, std=(w ** -0.5) * ((2 * n) ** -0.5))
            nn.init.normal_(blk.mlp.c_fc.weight, std=(2 * w) ** -0.5)
            nn.init.normal_(blk.mlp.c_proj.weight, std=(w ** -0.5) * ((2 * n) ** -0.5))
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    @property
    def dtype(self):
        return self.positional_embedding.dtype

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # a full CLIP checkpoint also carries the (unused) visual tower
        sd = {k: v for k, v in state_dict.items() if not k.startswith("visual.")}
        self._proj_t = None
        return super().load_state_dict(sd, strict=strict, **kw)

    def _proj(self, device):
        if self._proj_t is None or self._proj_t.device != device:
            self._proj_t = self.text_projection.detach().t().contiguous()     # frozen: [embed, W] = nn.Linear layout
        return lifetime.note(self._proj_t)

    def set_gemm_pieces(self, pieces: int):
        """0 = fp32 MFMA GEMMs (default, the parity path); 2 or 3 = split-bf16 products for the frozen block weights
        (2: 3 cross terms, ~3e-6 relative error, 2.3x the fp32 GEMM rate; 3: 6 terms, fp32-level error, 1.45x)."""
        for blk in self.transformer.resblocks:
            blk.gemm_pieces = int(pieces)
        return self

    @staticmethod
    def _live_rows(text: torch.Tensor):
        """The blocks are causal and only the EOT row is read out (clip/model.py:324-330,350-352), so rows behind a
        sequence's EOT cannot influence the result: run the tower on each sequence's prefix [0, eot] only.
        Returns (lengths, flat indices of the live rows in [P * ctx], index of every sequence's EOT row among them)."""
        P, ctx = text.shape
        eot = text.argmax(dim=-1).cpu().numpy().astype(np.int64)          # one small device->host copy per call
        lengths = (eot + 1).tolist()
        starts = np.arange(P, dtype=np.int64) * ctx
        live = np.concatenate([np.arange(s0, s0 + n, dtype=np.int64) for s0, n in zip(starts, lengths)])
        last = np.cumsum(eot + 1) - 1
        dev = text.device
        return lengths, torch.from_numpy(live).to(dev), torch.from_numpy(last).to(dev)

    def _tower_static(self, x_full: torch.Tensor, text: torch.Tensor) -> torch.Tensor:
        """Fixed-shape form of `_tower` (static_rows): every block runs on all P * ctx rows under the causal mask and the
        EOT rows are gathered on the device - the reference's own formulation (clip/model.py:343-350).  Nothing depends
        on the token ids on the host (no device->host copy, launch sequence independent of the notes), so a training step
        through the tower can be captured in a hipGraph; the price is the rows behind each EOT (about 2x the rows at
        20-40-token notes), which is why the live-prefix form stays the default."""
        P, ctx = text.shape
        segs = AttnSegs.make([ctx] * P, [ctx] * P, x_full.device)
        x = x_full
        for blk in self.transformer.resblocks:
            x = blk.flat(x, segs)
        last = torch.arange(P, device=text.device) * ctx + text.argmax(dim=-1)
        eot = ops.layer_norm(x.index_select(0, last), self.ln_final.weight, self.ln_final.bias, self.ln_final.eps)
        return ops.linear_act(eot, self._proj(eot.device))

    def _tower(self, x_full: torch.Tensor, text: torch.Tensor) -> torch.Tensor:
        """x_full [P * ctx, W] (embeddings + positions) -> EOT features [P, embed_dim]."""
        if self.static_rows:
            return self._tower_static(x_full, text)
        lengths, live, last = self._live_rows(text)
        x = x_full if live.numel() == x_full.shape[0] else x_full.index_select(0, live)
        segs = AttnSegs.make(lengths, lengths, x.device)
        blocks = list(self.transformer.resblocks)
        for blk in blocks[:-1]:
            x = blk.flat(x, segs)
        # Last block: only its EOT rows are read out, so q, out_proj and the MLP run on those rows alone; k and v are
        # still needed from every row of the prefix (the EOT query attends to all of them: no mask left to apply).
        x_e = blocks[-1].flat_last_rows(x, last, AttnSegs.make([1] * len(lengths), lengths, x.device))
        # ln_final is row-wise: normalise the EOT rows only
        eot = ops.layer_norm(x_e, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps)
        return ops.linear_act(eot, self._proj(eot.device))

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """text int64 [P, ctx] -> [P, embed_dim]."""
        x = ops.embed_tokens(text, self.token_embedding.weight, self.positional_embedding)     # [P*ctx, W]
        return self._tower(x, text)

    def encode_embedded(self, prompts: torch.Tensor, text: torch.Tensor) -> torch.Tensor:
        """Differentiable tower for already-embedded prompts (learnable context, model/dim1/CLIP.py:54-60):
        prompts [P, ctx, W] = token embeddings with the context rows replaced (positional embedding NOT yet
        added), text int64 [P, ctx] only locates the EOT row.  Gradients flow to `prompts`; the weights stay frozen."""
        P, ctx, W = prompts.shape
        return self._tower((prompts + self.positional_embedding).reshape(P * ctx, W), text)


def build_text_model(state_dict: dict) -> CLIPText:
    """Text-side counterpart of clip/model.py:395-432 build_model: sizes are read off the state_dict."""
    embed_dim = state_dict["text_projection"].shape[1]
    ctx = state_dict["positional_embedding"].shape[0]
    vocab = state_dict["token_embedding.weight"].shape[0]
    width = state_dict["ln_final.weight"].shape[0]
    layers = len({k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")})
    m = CLIPText(embed_dim, ctx, vocab, width, width // 64, layers)
    m.load_state_dict(state_dict, strict=False)
    return m.eval()
