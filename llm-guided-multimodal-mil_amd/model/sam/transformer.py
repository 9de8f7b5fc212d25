"""Two-way (text <-> image) cross-attention transformer, MI355X path.

Same module tree / parameter names / forward signature as the reference's
model/sam/transformer.py (TwoWayTransformer :10-120, TwoWayAttentionBlock :236-309, Attention :395-450),
so state_dicts interchange; the arithmetic runs on the HIP library: projections on the fp32-MFMA GEMM
(residual adds fused in the out_proj epilogue), the attention cores on the "pool" kernel (text tokens
over the patches of a bag) or the "rows" kernel (patches over the few text tokens), LayerNorm and the
positional add as streaming kernels.  Internally every tensor is flat [rows, E] with per-bag segments,
so a batch of ragged bags is one launch sequence (the reference runs one bag per forward)."""
import os
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from ...segments import AttnSegs
from .common import MLPBlock


class Attention(nn.Module):
    """q/k/v projections to internal_dim = embedding_dim // downsample_rate, H-head attention, out projection."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)

    def flat(self, q, k, v, segs: AttnSegs, form: str, residual=None):
        """q [Tq, E], k/v [Tk, E] flat rows; form 'pool' (few queries, many keys) or 'rows'."""
        qp = ops.linear_act(q, self.q_proj.weight, self.q_proj.bias)
        kp = ops.linear_act(k, self.k_proj.weight, self.k_proj.bias)
        vp = ops.linear_act(v, self.v_proj.weight, self.v_proj.bias)
        core = ops.attention_pool if form == "pool" else ops.attention_rows
        o = core(qp, kp, vp, segs, self.num_heads)
        return ops.linear_act(o, self.out_proj.weight, self.out_proj.bias, "none", residual=residual)

    def one_token(self, q, keys, pe_table, segs: AttnSegs, residual=None, qp=None):
        """Token->image attention when every bag has ONE text token: the K / V projections over the patches are
        absorbed into H query vectors and H pooled key vectors (ops.one_token_attention) - an HBM-bound pass over
        the keys instead of two [N, E] x [E, I] GEMMs.  keys come WITHOUT positional encoding (added on the fly).
        Returns (attention output, keys alias): later uses of the keys must go through the alias so that their
        gradient is folded inside the pool's backward."""
        _zero_grad_params(self.k_proj.bias)
        o, keys_pass = ops.one_token_attention(q, keys, pe_table, segs, self.q_proj.weight, self.q_proj.bias,
                                               self.k_proj.weight, self.v_proj.weight, self.v_proj.bias, self.num_heads, qp=qp)
        return ops.linear_act(o, self.out_proj.weight, self.out_proj.bias, "none", residual=residual), keys_pass

    def multi_token_pool(self, q, keys, kin, segs: AttnSegs, residual=None):
        """Token->image attention with T > 1 text tokens per bag, K / V projections absorbed (ops.multi_token_pool_attention):
        q [B*T, E] queries (+pe), keys [R, E] (values), kin [R, E] = keys + pe (scores).  Returns (output, keys alias):
        later uses of the keys go through the alias so that all their gradients are folded inside one node."""
        _zero_grad_params(self.k_proj.bias)
        o, keys_pass = ops.multi_token_pool_attention(q, keys, kin, segs, self.q_proj.weight, self.q_proj.bias,
                                                      self.k_proj.weight, self.v_proj.weight, self.v_proj.bias, self.num_heads)
        return ops.linear_act(o, self.out_proj.weight, self.out_proj.bias, "none", residual=residual), keys_pass

    def multi_token_rows(self, kin, k_tok, v_tok, segs: AttnSegs, residual=None):
        """Image->token attention with T > 1 tokens per bag, q / out projections absorbed: returns out_proj(attn) + residual."""
        return ops.multi_token_rows_attention(kin, k_tok, v_tok, segs, self.q_proj.weight, self.q_proj.bias,
                                              self.k_proj.weight, self.k_proj.bias, self.v_proj.weight, self.v_proj.bias,
                                              self.out_proj.weight, self.out_proj.bias, self.num_heads, residual=residual)

    def forward(self, q, k, v):
        """Reference signature: q [B, Tq, E], k, v [B, Tk, E] -> [B, Tq, E]."""
        B, Tq, E = q.shape
        Tk = k.shape[1]
        segs = AttnSegs.make([Tq] * B, [Tk] * B, q.device)
        form = "pool" if (Tq <= 16 and Tk > 16) else "rows"
        return self.flat(q.reshape(B * Tq, E), k.reshape(B * Tk, E), v.reshape(B * Tk, E), segs, form).reshape(B, Tq, E)


def _zero_grad_params(*params):
    """Parameters a fast path leaves out of the graph although the reference's graph contains them with a
    mathematically zero gradient (q / k projections of a one-key softmax, k_proj.bias under any softmax): upstream their
    .grad is a dense zero tensor, so torch.optim.Adam still applies weight decay to them, whereas a parameter whose .grad
    is None is skipped.  optim.FlatAdam reads this mark to tell the two cases apart."""
    for p in params:
        if p is not None:
            p._mil_zero_grad = True


def one_token_ok(attn: "Attention", s_ti: AttnSegs, pe_table) -> bool:
    """The absorbed one-token kernels are built for E = 512, 8 heads and exactly one query per bag."""
    return (pe_table is not None and s_ti.Tq_max == 1 and min(s_ti.q_lengths, default=1) == 1
            and attn.embedding_dim == 512 and attn.num_heads == 8 and attn.internal_dim // attn.num_heads in (32, 64))


class _LN(nn.LayerNorm):
    def forward(self, x, tail_rows: int = 0, into_tail_of=None):
        return ops.layer_norm(x, self.weight, self.bias, self.eps, tail_rows, into_tail_of)


class TwoWayAttentionBlock(nn.Module):
    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation: str = "relu",
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False):
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = _LN(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = _LN(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = _LN(embedding_dim)
        self.norm4 = _LN(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe

    def flat(self, queries, keys, query_pe, keys_pe_fn, s_tt: AttnSegs, s_ti: AttnSegs, s_it: AttnSegs, pe_table=None,
             keys_tail_rows: int = 0, pending=None, defer: bool = False):
        """One block on flat rows (sam/transformer.py:278-309).  keys_pe_fn(keys) = keys + key_pe.  With pe_table
        given and exactly one text token per bag, both cross attentions take their one-token forms and keys + pe is
        never materialised.  pending / defer (one-token fused route only, see TwoWayTransformer.flat): the block's last
        LayerNorm(keys + row) is left to the next attention site's pool kernel - `defer` returns (queries, keys, (row, norm4))
        with the norm still owed, `pending` is such a pair owed on the incoming keys."""
        if self._one_token_fused_ok(queries, s_tt, s_ti, s_it, pe_table):
            return self._flat_one_token_fused(queries, keys, query_pe, s_ti, s_it, pe_table, keys_tail_rows, pending, defer)
        if pending is not None or defer:
            raise ValueError("pending / defer are only defined on the one-token fused route")
        if s_tt.Tk_max == 1 and min(s_tt.k_lengths, default=1) == 1:
            # One text token per bag: self-attention over a single key returns that key's value whatever q and k
            # are (softmax of one score = 1), so :282-287 reduce to out_proj(v_proj(queries)) and q_proj / k_proj
            # get exactly zero gradient, as they do upstream.
            a = self.self_attn
            _zero_grad_params(a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias)
            vp = ops.linear_act(queries, a.v_proj.weight, a.v_proj.bias)
            queries = ops.linear_act(vp, a.out_proj.weight, a.out_proj.bias, "none",
                                     residual=None if self.skip_first_layer_pe else queries)
        elif self.skip_first_layer_pe:                                          # :282-283 (replaces, no residual)
            queries = self.self_attn.flat(queries, queries, queries, s_tt, "rows")
        else:                                                                   # :285-287
            q = queries + query_pe
            queries = self.self_attn.flat(q, q, queries, s_tt, "rows", residual=queries)
        queries = self.norm1(queries)
        q = queries + query_pe                                                  # :291-295
        one_token = one_token_ok(self.cross_attn_token_to_image, s_ti, pe_table)
        a0 = self.cross_attn_token_to_image
        multi = (not one_token) and ops.multi_token_ok(a0.embedding_dim, a0.num_heads, s_ti.q_lengths) \
            and a0.internal_dim // a0.num_heads in (32, 64)
        if one_token:
            k = None
            att, keys = self.cross_attn_token_to_image.one_token(q, keys, pe_table, s_ti, residual=queries)
            queries = self.norm2(att)
        elif multi:
            k = keys_pe_fn(keys)
            att, keys = self.cross_attn_token_to_image.multi_token_pool(q, keys, k, s_ti, residual=queries)
            queries = self.norm2(att)
        else:
            k = keys_pe_fn(keys)
            queries = self.norm2(self.cross_attn_token_to_image.flat(q, k, keys, s_ti, "pool", residual=queries))
        queries = self.norm3(self.mlp(queries, residual=queries))               # :298-300
        if s_it.Tk_max == 1 and min(s_it.k_lengths, default=1) == 1:
            # One text token per bag: the softmax over a single key is exactly 1, every patch receives the same
            # out_proj(v_proj(token)) and q_proj / k_proj get exactly zero gradient (as upstream).  Skips two
            # [N, 512] projections and the attention core; bit-for-bit the general path's result up to rounding.
            a = self.cross_attn_image_to_token
            _zero_grad_params(a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias)
            o = ops.linear_act(ops.linear_act(queries, a.v_proj.weight, a.v_proj.bias), a.out_proj.weight, a.out_proj.bias)
            keys = ops.layer_norm_bag_row(keys, o, s_it, self.norm4.weight, self.norm4.bias, self.norm4.eps, keys_tail_rows)
            return queries, keys                                                # (queries + query_pe of :303 has no reader here)
        q = queries + query_pe                                                  # :303-307
        if multi:
            keys = self.norm4(self.cross_attn_image_to_token.multi_token_rows(k, q, queries, s_it, residual=keys), keys_tail_rows)
        else:
            if k is None:
                k = keys_pe_fn(keys)
            keys = self.norm4(self.cross_attn_image_to_token.flat(k, q, queries, s_it, "rows", residual=keys), keys_tail_rows)
        return queries, keys

    # ------------------------------------------------------------------ one text token per bag: the token stream in fused links
    def _one_token_fused_ok(self, queries, s_tt, s_ti, s_it, pe_table) -> bool:
        a, t2i, i2t = self.self_attn, self.cross_attn_token_to_image, self.cross_attn_image_to_token
        return (s_tt.Tk_max == 1 and min(s_tt.k_lengths, default=1) == 1 and one_token_ok(t2i, s_ti, pe_table)
                and s_it.Tk_max == 1 and min(s_it.k_lengths, default=1) == 1
                and ops.lin_ln_lin_ok(queries, a.out_proj.weight, self.norm1.weight, t2i.q_proj.weight)
                and t2i.out_proj.weight.shape == (512, t2i.internal_dim) and t2i.internal_dim % 16 == 0
                and self.mlp.act == "relu" and self.mlp.lin1.weight.shape[1] == 512 and self.mlp.lin2.weight.shape[0] == 512
                and i2t.v_proj.weight.shape[1] == 512 and all(n.eps == self.norm1.eps for n in (self.norm2, self.norm3)))

    def _flat_one_token_fused(self, queries, keys, query_pe, s_ti, s_it, pe_table, keys_tail_rows, pending=None,
                              defer: bool = False):
        """The block for ONE text token per bag (the reference's `CI_prompt_version='single'`) with the token stream in three
        fused links P -> LayerNorm -> C (ops.lin_ln_lin: the norm rides in the operand load of the layer behind it, its
        backward in the operand load of the layer in front of it, gradient sums of two-consumer tensors inside the kernels):
          self-attention over one key = out_proj(v_proj(q)) (:282-287)  -> norm1 -> q_proj of the token->image attention (:291-295)
          absorbed pool + value projection (one node)                   -> its out_proj + residual -> norm2 -> mlp.lin1 (:298)
          mlp.lin2 + residual -> norm3 -> v_proj of the image->token attention (:303-307), out_proj, then LayerNorm(keys + row).
        Same arithmetic as the op-by-op route up to summation order; 11 launches less per block and step."""
        a, t2i, i2t, mlp = self.self_attn, self.cross_attn_token_to_image, self.cross_attn_image_to_token, self.mlp
        _zero_grad_params(a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias, t2i.k_proj.bias,
                          i2t.q_proj.weight, i2t.q_proj.bias, i2t.k_proj.weight, i2t.k_proj.bias)
        eps = self.norm1.eps
        # two consumers of the incoming queries: their gradients meet in the producer's backward kernel (ops.fan_out)
        qa, qb = (queries, None) if self.skip_first_layer_pe else ops.fan_out(queries, 2)
        vp = ops.linear_act(qa, a.v_proj.weight, a.v_proj.bias)
        qp, q1 = ops.lin_ln_lin(vp, a.out_proj.weight, a.out_proj.bias, qb,
                                self.norm1.weight, self.norm1.bias, eps, query_pe, t2i.q_proj.weight, t2i.q_proj.bias)
        if pending is not None:
            # the LayerNorm(keys + row) the block in front left to this pool (its kernel makes the keys it reads)
            prow, pn = pending
            o, keys = ops.lnbr_one_token_attention(keys, prow, pn.weight, pn.bias, pn.eps, pe_table, s_ti, t2i.k_proj.weight,
                                                   t2i.v_proj.weight, t2i.v_proj.bias, t2i.num_heads, qp)
        else:
            o, keys = ops.one_token_attention(None, keys, pe_table, s_ti, t2i.q_proj.weight, t2i.q_proj.bias, t2i.k_proj.weight,
                                              t2i.v_proj.weight, t2i.v_proj.bias, t2i.num_heads, qp=qp)
        h, q2 = ops.lin_ln_lin(o, t2i.out_proj.weight, t2i.out_proj.bias, q1, self.norm2.weight, self.norm2.bias, eps, None,
                               mlp.lin1.weight, mlp.lin1.bias, "relu")
        vp3, q3 = ops.lin_ln_lin(h, mlp.lin2.weight, mlp.lin2.bias, q2, self.norm3.weight, self.norm3.bias, eps, None,
                                 i2t.v_proj.weight, i2t.v_proj.bias)
        row = ops.linear_act(vp3, i2t.out_proj.weight, i2t.out_proj.bias)
        if defer:
            return q3, keys, (row, self.norm4)
        keys = ops.layer_norm_bag_row(keys, row, s_it, self.norm4.weight, self.norm4.bias, self.norm4.eps, keys_tail_rows)
        return q3, keys

    def forward(self, queries, keys, query_pe, key_pe):
        B, T, E = queries.shape
        N = keys.shape[1]
        dev = queries.device
        s_tt, s_ti, s_it = (AttnSegs.make([T] * B, [T] * B, dev), AttnSegs.make([T] * B, [N] * B, dev),
                            AttnSegs.make([N] * B, [T] * B, dev))
        kpe = key_pe.reshape(B * N, E)
        q, k = self.flat(queries.reshape(B * T, E), keys.reshape(B * N, E), query_pe.reshape(B * T, E),
                         lambda kk: kk + kpe, s_tt, s_ti, s_it)
        return q.reshape(B, T, E), k.reshape(B, N, E)


class TwoWayTransformer(nn.Module):
    def __init__(self, args, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int, activation: str = "relu",
                 attention_downsample_rate: int = 2):
        super().__init__()
        self.args = args
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate,
                                 skip_first_layer_pe=(i == 0)) for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = _LN(embedding_dim)

    def flat(self, image, point, pe_table, n_lengths, t_lengths, keys_tail_rows: int = 0,
             segs=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """image [sum N_b, E] patch tokens, point [sum T_b, E] text tokens, pe_table [>= max N_b, E].
        Returns (queries [sum T_b, E], keys [sum N_b, E])  (sam/transformer.py:100-120).  keys_tail_rows: the returned
        keys are allocated with room for that many more rows behind them (ops.append_rows).  segs: (s_tt, s_ti, s_it) of a
        segments.FusionBucket - bag lengths on the device, image = the bucket's `cap` rows - instead of host lengths."""
        dev = image.device
        if segs is not None:
            s_tt, s_ti, s_it = segs
        else:
            s_tt = AttnSegs.make(t_lengths, t_lengths, dev)
            s_ti = AttnSegs.make(t_lengths, n_lengths, dev)
            s_it = AttnSegs.make(n_lengths, t_lengths, dev)
        keys_pe = lambda kk: ops.add_pe(kk, pe_table, s_ti.k_bag, s_ti.k_off)      # noqa: E731
        # `point` feeds the first block's queries, every block's query_pe and the final attention: one handle each
        pa = ops.fan_out(point, len(self.layers) + 2)
        queries, keys = pa[0], image
        # One text token per bag: LayerNorm(keys + row), the last op of a block, is left to the pool kernel of the NEXT
        # attention site over those keys (ops.lnbr_one_token_attention) wherever both sides take their one-token forms.
        fuse = os.environ.get("MIL_FUSE_LNBR", "1") != "0"
        fa = self.final_attn_token_to_image
        pending = None
        for li, layer in enumerate(self.layers):
            last = li == len(self.layers) - 1
            nxt = fa if last else self.layers[li + 1].cross_attn_token_to_image
            defer = (fuse and layer._one_token_fused_ok(queries, s_tt, s_ti, s_it, pe_table)
                     and (one_token_ok(fa, s_ti, pe_table) if last
                          else self.layers[li + 1]._one_token_fused_ok(queries, s_tt, s_ti, s_it, pe_table))
                     and ops.lnbr_one_token_ok(keys, None, layer.norm4.weight, layer.norm4.bias, nxt.k_proj.weight,
                                               nxt.v_proj.weight, nxt.v_proj.bias, nxt.num_heads))
            out = layer.flat(queries, keys, pa[1 + li], keys_pe, s_tt, s_ti, s_it, pe_table, keys_tail_rows if last else 0,
                             pending=pending, defer=defer)
            queries, keys, pending = out if defer else (out[0], out[1], None)
        qf, qr = ops.fan_out(queries, 2)                                         # :114-118: q = queries + point; residual
        if pending is not None or one_token_ok(fa, s_ti, pe_table):
            # q_proj(queries + point): the sum is formed while the projection stages its operand
            qp = ops.linear_act(qf, fa.q_proj.weight, fa.q_proj.bias, x2=pa[-1])
            if pending is not None:
                prow, pn = pending
                _zero_grad_params(fa.k_proj.bias)
                o, keys = ops.lnbr_one_token_attention(keys, prow, pn.weight, pn.bias, pn.eps, pe_table, s_ti, fa.k_proj.weight,
                                                       fa.v_proj.weight, fa.v_proj.bias, fa.num_heads, qp, keys_tail_rows)
                out = ops.linear_act(o, fa.out_proj.weight, fa.out_proj.bias, "none", residual=qr)
            else:
                out, keys = fa.one_token(None, keys, pe_table, s_ti, residual=qr, qp=qp)
        elif ops.multi_token_ok(self.embedding_dim, self.num_heads, s_ti.q_lengths):
            q = qf + pa[-1]
            out, keys = fa.multi_token_pool(q, keys, keys_pe(keys), s_ti, residual=qr)
        else:
            q = qf + pa[-1]
            out = fa.flat(q, keys_pe(keys), keys, s_ti, "pool", residual=qr)
        # the returned text tokens are the rows the multi-modal bag appends behind the patch tokens (model/aggregator.py:192):
        # when exactly that many rows were reserved behind `keys`, the last LayerNorm writes them there (no copy launch)
        return self.norm_final_attn(out, into_tail_of=keys if keys_tail_rows == out.shape[0] else None), keys

    def forward(self, image_embedding, image_pe, point_embedding):
        """Reference signature (sam/transformer.py:58-63): image_embedding [B, N, E], image_pe [B or 1, N, E],
        point_embedding [B, T, E] -> (queries [B, T, E], keys [B, N, E]).  A 5-D image_embedding is a CT feature map
        [B, E, D, h, w] (:78-98): it becomes D tokens (mean over h, w; resnetMC3_18) or D*h*w tokens (medicalNet)."""
        if getattr(self.args, "alignment_base", "CI") == "CT":
            # :78-86: the CT map arrives as POINT embedding and its D tokens are the queries; the image side stays [B, N, E].
            # D x 8 heads exceeds every few-token fast path: self-attention over D tokens (<= 96: the whole-sequence kernel), D
            # queries over N keys and N queries over D keys go through the general rows kernels (mil_attn_rows_fwd /
            # mil_attn_rows_bwd_general) - no shipped run takes this branch, so it is correct rather than fast.
            if point_embedding.dim() == 5:
                Bc = point_embedding.shape[0]
                rows, Tn = ops.ct_map_tokens(point_embedding, getattr(self.args, "model_CT", "resnetMC3_18"))
                point_embedding = rows.view(Bc, Tn, rows.shape[1])
        elif image_embedding.dim() == 5:
            Bc = image_embedding.shape[0]
            rows, Tn = ops.ct_map_tokens(image_embedding, getattr(self.args, "model_CT", "resnetMC3_18"))
            image_embedding = rows.view(Bc, Tn, rows.shape[1])
        if image_embedding.dim() != 3:
            raise ValueError("image_embedding must be [B, N, E] or a 5-D CT map [B, E, D, h, w]")
        B, N, E = image_embedding.shape
        T = point_embedding.shape[1]
        pe = image_pe.reshape(-1, E)[:N].contiguous()
        q, k = self.flat(image_embedding.reshape(B * N, E), point_embedding.reshape(B * T, E), pe, [N] * B, [T] * B)
        return q.reshape(B, T, E), k.reshape(B, N, E)
