"""Host-side operators over the C ABI (include/mil_hip.h).

Two layers:
  * plain functions, one per C entry point: torch tensors in, torch tensors out, all work
    enqueued on the current HIP stream, no host sync;
  * ``torch.autograd.Function`` wrappers (``gated_attention_pool``, ``head_sigmoid``) so the
    ``aggregator`` module composes with autograd/DDP exactly like the reference's
    ABMIL (model/dim1/ABMIL.py:47-64) and ``fc`` + sigmoid (model/aggregator.py:128-131,200).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib, deferred
from .bags import BagLayout

GATE_D = 192


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.MilHipError(f"{name}: expected a tensor on the MI355X (cuda) device; there is no CPU path")
    if t.dtype != torch.float32:
        raise _lib.MilHipError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- plain operators
X_DROP_P, X_DROP_SCALE = 0.5, 2.0            # ABMIL.py:26,49
M_DROP_P, M_DROP_SCALE = 0.25, 1.0 / 0.75    # aggregator.py:129


def dropout_keep_bits(rows: int, cols: int, p_drop: float, seed: int, offset: int, device, out=None, offset_dev=None):
    """uint32-packed keep mask [rows, cols // 32] (stored as int32) from Philox4x32-10 (csrc/dropout.hip)."""
    if cols % 32:
        raise _lib.MilHipError("dropout_keep_bits: cols must be a multiple of 32")
    if out is None:
        out = torch.empty((rows, cols // 32), device=device, dtype=torch.int32)
    rc = _lib.lib().mil_dropout_keep_bits(_p(out), rows, cols, float(p_drop), int(seed) & (2 ** 64 - 1),
                                          int(offset) & (2 ** 64 - 1), _p(offset_dev), _stream())
    _lib.check(rc, "mil_dropout_keep_bits")
    return out


def dropout_keep_bits_pair(rows: int, bags: int, cols: int, seed: int, mseed: int, counter, done):
    """(xbits [rows, cols/32] at p = 0.5, mbits [bags, cols/32] at p = 0.25) drawn at stream position counter[0] in ONE launch
    that also advances the counter (mil_dropout_keep_bits_pair); rows == 0 / bags == 0: that tensor is None."""
    dev = counter.device
    xb = torch.empty((rows, cols // 32), device=dev, dtype=torch.int32) if rows else None
    mb = torch.empty((bags, cols // 32), device=dev, dtype=torch.int32) if bags else None
    m64 = 2 ** 64 - 1
    # mdelta = 1: the head's words at the position the separate launch drew them (behind the counter's increment)
    rc = _lib.lib().mil_dropout_keep_bits_pair(_p(xb), rows, _p(mb), bags, cols, int(seed) & m64, int(mseed) & m64, 0,
                                               _p(counter), _p(counter), _p(done), 1, _stream())
    _lib.check(rc, "mil_dropout_keep_bits_pair")
    return xb, mb


def counter_add(counter, v: int = 1):
    """counter[0] += v on the current stream (device int32)."""
    rc = _lib.lib().mil_counter_add(_p(counter), int(v), _stream())
    _lib.check(rc, "mil_counter_add")


def dropout_apply_bits(t, bits, scale: float):
    """t = keep ? t * scale : 0 in place."""
    rows, cols = t.shape
    rc = _lib.lib().mil_dropout_apply_bits(_p(t), _p(bits), rows, cols, float(scale), _stream())
    _lib.check(rc, "mil_dropout_apply_bits")
    return t


class _DropoutBits(torch.autograd.Function):
    """y = keep ? t / (1 - p) : 0 through a packed keep-bit tensor; the backward reads the same bits."""

    @staticmethod
    def forward(ctx, t, bits, scale: float):
        ctx.save_for_backward(bits)
        ctx.scale = scale
        return dropout_apply_bits(t.detach().clone().contiguous(), bits, scale)

    @staticmethod
    def backward(ctx, dy):
        (bits,) = ctx.saved_tensors
        return dropout_apply_bits(dy.detach().clone().contiguous(), bits, ctx.scale), None, None


def dropout_bits(t, bits, scale: float):
    """Differentiable form of dropout_apply_bits (out of place): the head's Dropout(.25) of the autograd route draws the same
    Philox keep words as the fused tail (csrc/dropout.hip) instead of torch's generator."""
    return _DropoutBits.apply(t, bits, scale)


def gate_scores_fwd(x, Wv, bv, Wu, bu, w, b, save_gates: bool = True, xbits=None, xscale: float = 1.0):
    """scores [R], gates [R, 384] (or None).  ABMIL.py:52-54.  xbits: keep bits of the patch dropout (train mode)."""
    x = _f32c(x, "x")
    R, L = x.shape
    scores = torch.empty(R, device=x.device, dtype=torch.float32)
    gates = torch.empty((R, 2 * GATE_D), device=x.device, dtype=torch.float32) if save_gates else None
    rc = _lib.lib().mil_gate_scores_fwd(_p(x), _p(_f32c(Wv, "Wv")), _p(_f32c(bv, "bv")), _p(_f32c(Wu, "Wu")),
                                        _p(_f32c(bu, "bu")), _p(_f32c(w, "w")), _p(_f32c(b, "b")), _p(scores),
                                        _p(gates), R, L, Wv.shape[0], _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_gate_scores_fwd")
    return scores, gates


def attn_pool_fwd(x, scores, layout: BagLayout, xbits=None, xscale: float = 1.0):
    """M [B, L], lse [B].  ABMIL.py:56-59 per bag."""
    x = _f32c(x, "x")
    R, L = x.shape
    if R != layout.R:
        raise _lib.MilHipError(f"attn_pool_fwd: x has {R} rows but the bag layout covers {layout.R}")
    partials = torch.empty(layout.T * (L + 2), device=x.device, dtype=torch.float32)
    M = torch.empty((layout.B, L), device=x.device, dtype=torch.float32)
    lse = torch.empty(layout.B, device=x.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_fwd(_p(x), _p(scores), _p(layout.tile_map), _p(layout.bag_tile_off), layout.T,
                                      layout.B, L, _p(partials), _p(M), _p(lse), _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_attn_pool_fwd")
    return M, lse


def attn_pool_partial(x, scores, layout: BagLayout, xbits=None, xscale: float = 1.0):
    """Tile partials only ([T*L] weighted sums then [T*2] (max, sum) pairs); merged by pool_merge_head."""
    x = _f32c(x, "x")
    R, L = x.shape
    if R != layout.R:
        raise _lib.MilHipError(f"attn_pool_partial: x has {R} rows but the bag layout covers {layout.R}")
    partials = torch.empty(layout.T * (L + 2), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_partial(_p(x), _p(scores), _p(layout.tile_map), layout.T, L, _p(partials), _p(xbits),
                                          float(xscale), _stream())
    _lib.check(rc, "mil_attn_pool_partial")
    return partials


def attn_pool_partial_h(x, scores, layout: BagLayout, Wf, xbits=None, xscale: float = 1.0, mbits=None, mscale: float = 1.0):
    """Tile partials plus hrow [R, C] = x Wf^T (head projection of every patch; lets the backward skip x)."""
    x = _f32c(x, "x")
    R, L = x.shape
    C = Wf.shape[0]
    partials = torch.empty(layout.T * (L + 2), device=x.device, dtype=torch.float32)
    hrow = torch.empty((R, C), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_partial_h(_p(x), _p(scores), _p(layout.tile_map), layout.T, L, _p(partials),
                                            _p(_f32c(Wf, "Wf")), C, _p(hrow), _p(xbits), float(xscale), _p(mbits),
                                            float(mscale), _stream())
    _lib.check(rc, "mil_attn_pool_partial_h")
    return partials, hrow


def attn_pool_bwd_from_h(scores, lse, hrow, dz, cdot, layout: BagLayout):
    ds = torch.empty(scores.shape[0], device=scores.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_bwd_from_h(_p(scores), _p(lse), _p(hrow), _p(dz), _p(cdot), _p(layout.tile_map),
                                             layout.T, hrow.shape[1], _p(ds), _stream())
    _lib.check(rc, "mil_attn_pool_bwd_from_h")
    return ds


def pool_merge_head(partials, layout: BagLayout, L: int, Wf, bf, y=None, scale: float = 1.0, scores=None, hrow=None,
                    mbits=None, mscale: float = 1.0, loss_kind: int = 0):
    """Fused per-bag tail: returns dict(M, lse, logits, prob[, loss_bag, dz, dM, cdot[, ds]]); with labels it also
    produces each bag's scaled BCE loss and the head's backward inputs for the pool, and with the forward's head
    projections `hrow` (attn_pool_partial_h) the score gradient ds of every row as well."""
    B, C, dev = layout.B, Wf.shape[0], partials.device
    out = dict(M=torch.empty((B, L), device=dev), lse=torch.empty(B, device=dev),
               logits=torch.empty((B, C), device=dev), prob=torch.empty((B, C), device=dev))
    if y is not None:
        out.update(dz=torch.empty((B, C), device=dev), dM=torch.empty((B, L), device=dev),
                   cdot=torch.empty(B, device=dev), loss_bag=torch.empty(B, device=dev))
        if hrow is not None and scores is not None:
            # capacity bucket (segments.FusionBucket): rows outside every tile are padding and must read ds = 0 - the bucket
            # owns the buffer and its refresh() launch zeroes those rows
            dsb = getattr(layout, "ds_buffer", None)
            out["ds"] = dsb if (dsb is not None and dsb.shape[0] == scores.shape[0]) else \
                (torch.zeros if getattr(layout, "device_lengths", False) else torch.empty)(scores.shape[0], device=dev)
    if mbits is not None:
        out["Mdrop"] = torch.empty((B, L), device=dev)
    # long bags (one ragged bag per step): the tail spreads over many workgroups through a small workspace
    ws = torch.empty(_lib.lib().mil_pool_tail_workspace_floats(B), device=dev) if ("ds" in out and layout.T >= 64 * B and B <= 8) else None
    rc = _lib.lib().mil_pool_merge_head_ws(_p(partials), _p(layout.bag_tile_off), layout.T, B, L, _p(_f32c(Wf, "Wf")),
                                           _p(_f32c(bf, "bf")), C, _p(y), float(scale), _p(out["M"]), _p(out["lse"]),
                                           _p(out["logits"]), _p(out["prob"]), _p(out.get("loss_bag")), _p(out.get("dz")),
                                           _p(out.get("dM")), _p(out.get("cdot")),
                                           _p(layout.tile_map) if "ds" in out else None, _p(scores) if "ds" in out else None,
                                           _p(hrow) if "ds" in out else None, _p(out.get("ds")), _p(mbits), float(mscale),
                                           _p(out.get("Mdrop")), int(loss_kind), _p(ws), _stream())
    _lib.check(rc, "mil_pool_merge_head_ws")
    return out


def head_bwd_params(dz, M, dWf, dbf, loss_bag=None, loss_out=None):
    B, L = M.shape
    rc = _lib.lib().mil_head_bwd_params(_p(dz), _p(M), _p(dWf), _p(dbf), B, L, dz.shape[1], _p(loss_bag), _p(loss_out),
                                        _stream())
    _lib.check(rc, "mil_head_bwd_params")


def head_fwd(M, Wf, bf):
    """logits z [B, C], p = sigmoid(z).  aggregator.py:128-131,200 (eval)."""
    M = _f32c(M, "M")
    B, L = M.shape
    C = Wf.shape[0]
    z = torch.empty((B, C), device=M.device, dtype=torch.float32)
    p = torch.empty_like(z)
    rc = _lib.lib().mil_head_fwd(_p(M), _p(_f32c(Wf, "Wf")), _p(_f32c(bf, "bf")), _p(z), _p(p), B, L, C, _stream())
    _lib.check(rc, "mil_head_fwd")
    return z, p


def bce_fwd_bwd(p, y, scale: float, loss_sum: Optional[torch.Tensor] = None):
    """Adds sum(BCE) * scale into loss_sum [1] and returns (loss_sum, dz = (p - y) * scale)."""
    B, C = p.shape
    if loss_sum is None:
        loss_sum = torch.zeros(1, device=p.device, dtype=torch.float32)
    dz = torch.empty_like(p)
    rc = _lib.lib().mil_bce_fwd_bwd(_p(_f32c(p, "p")), _p(_f32c(y, "y")), _p(loss_sum), _p(dz), B, C, float(scale),
                                    _stream())
    _lib.check(rc, "mil_bce_fwd_bwd")
    return loss_sum, dz


def head_bwd(dz_or_dp, p, M, Wf):
    """(dM [B, L], dWf [C, L], dbf [C], cdot [B]).  If p is given the first argument is dL/dp."""
    B, L = M.shape
    C = Wf.shape[0]
    dM = torch.empty_like(M)
    dWf = torch.empty((C, L), device=M.device, dtype=torch.float32)
    dbf = torch.empty(C, device=M.device, dtype=torch.float32)
    cdot = torch.empty(B, device=M.device, dtype=torch.float32)
    rc = _lib.lib().mil_head_bwd(_p(_f32c(dz_or_dp, "dz")), _p(p), _p(_f32c(M, "M")), _p(_f32c(Wf, "Wf")), _p(dM),
                                 _p(dWf), _p(dbf), _p(cdot), B, L, C, _stream())
    _lib.check(rc, "mil_head_bwd")
    return dM, dWf, dbf, cdot


def rowdot(a, c):
    B, L = a.shape
    out = torch.empty(B, device=a.device, dtype=torch.float32)
    rc = _lib.lib().mil_rowdot(_p(_f32c(a, "a")), _p(_f32c(c, "c")), _p(out), B, L, _stream())
    _lib.check(rc, "mil_rowdot")
    return out


def attn_pool_bwd(x, scores, lse, dM, cdot, layout: BagLayout, want_dx: bool, xbits=None, xscale: float = 1.0):
    """ds [R] and, if requested, the pool term of dx ([R, L] = A_i dM)."""
    x = _f32c(x, "x")
    R, L = x.shape
    ds = torch.empty(R, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if want_dx else None
    rc = _lib.lib().mil_attn_pool_bwd(_p(x), _p(scores), _p(lse), _p(_f32c(dM, "dM")), _p(cdot), _p(layout.tile_map),
                                      layout.T, L, _p(ds), _p(dx), _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_attn_pool_bwd")
    return ds, dx


def gate_bwd_params(x, gates, ds, w, dWv, dbv, dWu, dbu, dw, db, accumulate: bool = False,
                    workspace: Optional[torch.Tensor] = None, xbits=None, xscale: float = 1.0):
    x = _f32c(x, "x")
    R, L = x.shape
    need = _lib.lib().mil_gate_bwd_workspace_floats(R, L)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.float32)
    rc = _lib.lib().mil_gate_bwd_params(_p(x), _p(gates), _p(ds), _p(_f32c(w, "w")), R, L, GATE_D, _p(workspace),
                                        workspace.numel(), _p(dWv), _p(dbv), _p(dWu), _p(dbu), _p(dw), _p(db),
                                        1 if accumulate else 0, _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_gate_bwd_params")
    return workspace


def gate_bwd_params_head(x, gates, ds, w, dWv, dbv, dWu, dbu, dw, db, dz, M, dWf, dbf, loss_bag=None, loss_out=None,
                         workspace: Optional[torch.Tensor] = None, xbits=None, xscale: float = 1.0):
    """gate_bwd_params + head_bwd_params in two launches instead of three: the head's parameter gradients are computed by
    workgroups appended to the reduce launch (mil_gate_bwd_params_head)."""
    x = _f32c(x, "x")
    R, L = x.shape
    need = _lib.lib().mil_gate_bwd_workspace_floats(R, L)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.float32)
    B, C = dz.shape
    if M.shape[1] != L:
        raise _lib.MilHipError("gate_bwd_params_head: the bag embeddings must have the gate's input width")
    rc = _lib.lib().mil_gate_bwd_params_head(_p(x), _p(gates), _p(ds), _p(_f32c(w, "w")), R, L, GATE_D, _p(workspace),
                                             workspace.numel(), _p(dWv), _p(dbv), _p(dWu), _p(dbu), _p(dw), _p(db), 0,
                                             _p(dz), _p(M), _p(dWf), _p(dbf), B, C, _p(loss_bag), _p(loss_out), _p(xbits),
                                             float(xscale), _stream())
    _lib.check(rc, "mil_gate_bwd_params_head")
    return workspace


def gate_bwd_input(gates, ds, w, Wv, Wu, dx, xbits=None, xscale: float = 1.0):
    R, L = dx.shape
    rc = _lib.lib().mil_gate_bwd_input(_p(gates), _p(ds), _p(_f32c(w, "w")), _p(_f32c(Wv, "Wv")), _p(_f32c(Wu, "Wu")),
                                       R, L, GATE_D, _p(dx), _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_gate_bwd_input")
    return dx


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
              weight_decay: float = 1e-7, grad_scale: float = 1.0):
    rc = _lib.lib().mil_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), int(step), lr,
                                  betas[0], betas[1], eps, weight_decay, grad_scale, _stream())
    _lib.check(rc, "mil_adam_step")


def adam_step_counted(param, grad, exp_avg, exp_avg_sq, step_counter, lr: float = 1e-5, betas=(0.9, 0.999),
                      eps: float = 1e-8, weight_decay: float = 1e-7, grad_scale: float = 1.0):
    """Adam with the step number in a device int32 (incremented by the call): same launches every step."""
    rc = _lib.lib().mil_adam_step_counted(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(),
                                          _p(step_counter), lr, betas[0], betas[1], eps, weight_decay, grad_scale,
                                          _stream())
    _lib.check(rc, "mil_adam_step_counted")


def adam_step_counted_noinc(param, grad, exp_avg, exp_avg_sq, step_counter, lr: float = 1e-5, betas=(0.9, 0.999),
                            eps: float = 1e-8, weight_decay: float = 1e-7, grad_scale: float = 1.0):
    """One segment of a counted Adam step; the caller advances the counter once (counter_add)."""
    rc = _lib.lib().mil_adam_step_counted_noinc(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(),
                                                _p(step_counter), lr, betas[0], betas[1], eps, weight_decay, grad_scale,
                                                _stream())
    _lib.check(rc, "mil_adam_step_counted_noinc")


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, step_counter, lr_dev, betas=(0.9, 0.999), eps: float = 1e-8,
                  weight_decay: float = 1e-7, grad_scale: float = 1.0, inc: bool = True):
    """Counted Adam with the learning rate in device memory too (lr_dev [1]): a captured step follows the schedule."""
    rc = _lib.lib().mil_adam_step_dev(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(step_counter),
                                      _p(lr_dev), betas[0], betas[1], eps, weight_decay, grad_scale, 1 if inc else 0, _stream())
    _lib.check(rc, "mil_adam_step_dev")


def adam_step_dev_segs(param, grad, exp_avg, exp_avg_sq, segs, step_counter, lr_dev, done_counter, betas=(0.9, 0.999),
                       eps: float = 1e-8, weight_decay: float = 1e-7, grad_scale: float = 1.0, inc: bool = True):
    """Adam over the ranges `segs` = [(begin, end), ...] of the flat buffers + the step-counter advance, one launch."""
    import ctypes
    n = len(segs)
    b = (ctypes.c_size_t * n)(*[int(a_) for a_, _ in segs])
    e = (ctypes.c_size_t * n)(*[int(b_) for _, b_ in segs])
    rc = _lib.lib().mil_adam_step_dev_segs(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), b, e, n, _p(step_counter), _p(lr_dev),
                                           _p(done_counter), betas[0], betas[1], eps, weight_decay, grad_scale, int(inc), _stream())
    _lib.check(rc, "mil_adam_step_dev_segs")


def sgd_step(param, grad, lr: float = 1e-3, weight_decay: float = 1e-7, grad_scale: float = 1.0):
    """torch.optim.SGD (no momentum, L2 weight decay) over a flat buffer, in place."""
    rc = _lib.lib().mil_sgd_step(_p(param), _p(grad), param.numel(), lr, weight_decay, grad_scale, _stream())
    _lib.check(rc, "mil_sgd_step")


# --------------------------------------------------------------------------- autograd wrappers
class _GatedAttentionPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Wv, bv, Wu, bu, w, b, layout: BagLayout, xbits=None):
        x = _f32c(x, "x")
        need_grad = any(ctx.needs_input_grad[:7])
        xs = X_DROP_SCALE if xbits is not None else 1.0
        scores, gates = gate_scores_fwd(x, Wv, bv, Wu, bu, w.reshape(-1), b, save_gates=need_grad, xbits=xbits, xscale=xs)
        M, lse = attn_pool_fwd(x, scores, layout, xbits=xbits, xscale=xs)
        ctx.layout, ctx.xbits, ctx.xs = layout, xbits, xs
        ctx.save_for_backward(x, Wv, Wu, w, scores, gates if gates is not None else torch.empty(0, device=x.device), lse, M)
        ctx.mark_non_differentiable(scores)
        return M, scores

    @staticmethod
    def backward(ctx, dM, _dscores):
        x, Wv, Wu, w, scores, gates, lse, M = ctx.saved_tensors
        dM = _f32c(dM, "dM")
        cdot = rowdot(M, dM)
        want_dx = ctx.needs_input_grad[0]
        xbits, xs = ctx.xbits, ctx.xs
        ds, dx = attn_pool_bwd(x, scores, lse, dM, cdot, ctx.layout, want_dx, xbits=xbits, xscale=xs)
        dWv = torch.empty_like(Wv)
        dWu = torch.empty_like(Wu)
        dbv = torch.empty(GATE_D, device=x.device, dtype=torch.float32)
        dbu = torch.empty_like(dbv)
        dw = torch.empty_like(dbv)
        db = torch.empty(1, device=x.device, dtype=torch.float32)
        wflat = w.reshape(-1)
        gate_bwd_params(x, gates, ds, wflat, dWv, dbv, dWu, dbu, dw, db, xbits=xbits, xscale=xs)
        if want_dx:
            gate_bwd_input(gates, ds, wflat, Wv, Wu, dx, xbits=xbits, xscale=xs)      # also applies the dropout backward
        return dx, dWv, dbv, dWu, dbu, dw.reshape(w.shape), db, None, None


def gated_attention_pool(x, Wv, bv, Wu, bu, w, b, layout: BagLayout, xbits=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """M [B, L] (differentiable) and the raw attention scores [R] (not differentiable).  xbits: keep bits of the
    patch dropout (train mode, ABMIL.py:49): the kernels read x through the mask, no dropped copy is made."""
    return _GatedAttentionPool.apply(x, Wv, bv, Wu, bu, w, b, layout, xbits)


def gate_bwd_input_pool(gates, ds, w, Wv, Wu, dx, scores, lse, row_bag, dM, xbits=None, xscale: float = 1.0):
    """dx = a_row dM[bag(row)] + dPre [Wv; Wu] written in ONE pass (mil_gate_bwd_input_pool): the attention pool's own input
    gradient is formed in the epilogue of the gate's input-gradient product, dx is never read."""
    R, L = dx.shape
    rc = _lib.lib().mil_gate_bwd_input_pool(_p(gates), _p(ds), _p(_f32c(w, "w")), _p(_f32c(Wv, "Wv")), _p(_f32c(Wu, "Wu")),
                                            R, L, GATE_D, _p(dx), _p(xbits), float(xscale), _p(scores), _p(lse), _p(row_bag),
                                            _p(dM), _stream())
    _lib.check(rc, "mil_gate_bwd_input_pool")
    return dx


class _GatedPoolHeadLoss(torch.autograd.Function):
    """ABMIL pool -> Dropout(.25) -> fc -> sigmoid -> BCELoss(mean) (or CE on the sigmoid outputs) as ONE autograd node
    (ABMIL.py:47-59 + aggregator.py:128-131,200 + train_ddp.py:95-99,323-324), on the fused per-bag tail of the image-only
    step: gate forward, pool partial pass with the head-projection by-product, one tail launch (merge, head, loss, dz, dM,
    ds) and the head's parameter gradients - 4 launches where the op-by-op route takes 17; the backward is the gate's
    weight-gradient product, its reduce, and ONE pass writing dx (pool term + gate term).
    Returns (loss [scalar], prob [B, C], logits [B, C], M [B, L]); only the loss is differentiable, and it must be the
    root of the backward pass (its incoming gradient is taken to be 1; scale through `scale`)."""

    _checked_unit_grad = False

    @staticmethod
    def forward(ctx, x, Wv, bv, Wu, bu, w, b, Wf, bf, y, layout: BagLayout, scale: float, loss_kind: int, xbits, mbits):
        x = _f32c(x, "x")
        L = x.shape[1]
        xs = X_DROP_SCALE if xbits is not None else 1.0
        ms = M_DROP_SCALE if mbits is not None else 1.0
        need_grad = any(ctx.needs_input_grad[:9])
        scores, gates = gate_scores_fwd(x, Wv, bv, Wu, bu, w.reshape(-1), b, save_gates=need_grad, xbits=xbits, xscale=xs)
        partials, hrow = attn_pool_partial_h(x, scores, layout, Wf, xbits, xs, mbits, ms)
        t = pool_merge_head(partials, layout, L, Wf, bf, _f32c(y, "y"), scale, scores, hrow, mbits, ms, loss_kind)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        dWf = (grad_slot(Wf) if need_grad else None)
        dbf = (grad_slot(bf) if need_grad else None)
        dWf = dWf if dWf is not None else torch.empty_like(Wf)
        dbf = dbf if dbf is not None else torch.empty_like(bf)
        head_bwd_params(t["dz"], t.get("Mdrop", t["M"]), dWf, dbf, t["loss_bag"], loss)
        ctx.layout, ctx.xbits, ctx.xs = layout, xbits, xs
        ctx.params = (Wv, bv, Wu, bu, w, b)
        ctx.save_for_backward(x, Wv, Wu, w, gates if gates is not None else torch.empty(0, device=x.device), scores,
                              t["lse"], t["ds"], t["dM"], dWf, dbf)
        for o in (t["prob"], t["logits"], t["M"]):
            ctx.mark_non_differentiable(o)
        ctx.set_materialize_grads(False)        # no zero tensors for the three non-differentiable outputs
        return loss.reshape(()), t["prob"], t["logits"], t["M"]

    @staticmethod
    def backward(ctx, dloss, _dp, _dz, _dM):
        x, Wv, Wu, w, gates, scores, lse, ds, dM, dWf, dbf = ctx.saved_tensors
        if not _GatedPoolHeadLoss._checked_unit_grad and not torch.cuda.is_current_stream_capturing():
            _GatedPoolHeadLoss._checked_unit_grad = True
            if abs(float(dloss) - 1.0) > 1e-6:
                raise _lib.MilHipError("fused pool+head+loss: the loss must be the root of backward() (incoming gradient 1); "
                                       "fold any factor into `scale`")
        pW = ctx.params
        outs = []
        for prm in pW:
            slot = grad_slot(prm)
            outs.append(slot if slot is not None else torch.empty(prm.shape, device=x.device, dtype=torch.float32))
        dWv, dbv, dWu, dbu, dw, db = outs
        wflat = w.reshape(-1)
        xbits, xs = ctx.xbits, ctx.xs
        gate_bwd_params(x, gates, ds, wflat, dWv, dbv, dWu, dbu, dw, db, xbits=xbits, xscale=xs)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gate_bwd_input_pool(gates, ds, wflat, Wv, Wu, dx, scores, lse, ctx.layout.row_bag(), dM, xbits=xbits, xscale=xs)
        return dx, dWv, dbv, dWu, dbu, dw, db, dWf, dbf, None, None, None, None, None, None


def gated_pool_head_loss(x, Wv, bv, Wu, bu, w, b, Wf, bf, y, layout: BagLayout, scale: float, loss_kind: int = 0,
                         xbits=None, mbits=None):
    """(loss, prob, logits, M) of the fused ABMIL + head + loss node (see _GatedPoolHeadLoss)."""
    return _GatedPoolHeadLoss.apply(x, Wv, bv, Wu, bu, w, b, Wf, bf, y, layout, float(scale), int(loss_kind), xbits, mbits)


class _HeadSigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M, Wf, bf):
        z, p = head_fwd(M, Wf, bf)
        ctx.save_for_backward(M, Wf, p)
        ctx.mark_non_differentiable(z)
        return p, z

    @staticmethod
    def backward(ctx, dp, _dz):
        M, Wf, p = ctx.saved_tensors
        dM, dWf, dbf, _ = head_bwd(_f32c(dp, "dp"), p, M, Wf)
        return dM, dWf, dbf


def head_sigmoid(M, Wf, bf):
    """(p = sigmoid(fc(M)) [B, C] differentiable, logits z [B, C])."""
    return _HeadSigmoid.apply(M, Wf, bf)


# --------------------------------------------------------------------------- generic fp32-MFMA linear (K3a)
ACT = {"none": 0, "tanh": 1, "relu": 2, "quickgelu": 3}


def gemm(A, a_mode: int, B, b_mode: int, M: int, N: int, K: int, out=None, bias=None, act: int = 0, residual=None,
         accumulate: bool = False, split_k: bool = True, rows_dev=None):
    """C[M,N] (+)= act(A_op . B_op + bias) + residual  (include/mil_hip.h: mil_gemm)."""
    A = _f32c(A, "A")
    B = _f32c(B, "B")
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    ws, nws = None, 0
    if split_k:
        nws = _lib.lib().mil_gemm_workspace_floats(M, N, K, a_mode)
        if nws:
            ws = torch.empty(nws, device=A.device, dtype=torch.float32)
    if rows_dev is not None:       # capacity bucket: true row count on the device (include/mil_hip.h: mil_gemm_rows)
        rc = _lib.lib().mil_gemm_rows(_p(A), A.stride(0), a_mode, _p(B), B.stride(0), b_mode, _p(out), out.stride(0), M, N, K,
                                      _p(bias), act, _p(residual), residual.stride(0) if residual is not None else 0,
                                      1 if accumulate else 0, _p(ws), nws, _p(rows_dev), _stream())
        _lib.check(rc, "mil_gemm_rows")
        return out
    rc = _lib.lib().mil_gemm(_p(A), A.stride(0), a_mode, _p(B), B.stride(0), b_mode, _p(out), out.stride(0), M, N, K,
                             _p(bias), act, _p(residual), residual.stride(0) if residual is not None else 0,
                             1 if accumulate else 0, _p(ws), nws, _stream())
    _lib.check(rc, "mil_gemm")
    return out


def gemm_aux(A, B, b_mode: int, M: int, N: int, K: int, aux, aux_mode: int, bias=None, act: int = 0, residual=None):
    """C = act(A . B_op + bias) + residual with one auxiliary [M, N] tensor handled in the epilogue
    (include/mil_hip.h: mil_gemm_aux; aux_mode 1 = store the pre-activation, 2 = multiply by QuickGELU'(aux))."""
    A, B = _f32c(A, "A"), _f32c(B, "B")
    out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    nws = _lib.lib().mil_gemm_workspace_floats(M, N, K, 0)
    ws = torch.empty(nws, device=A.device, dtype=torch.float32) if nws else None
    rc = _lib.lib().mil_gemm_aux(_p(A), A.stride(0), 0, _p(B), B.stride(0), b_mode, _p(out), out.stride(0), M, N, K,
                                 _p(bias), act, _p(residual), residual.stride(0) if residual is not None else 0, 0,
                                 _p(ws), nws, _p(aux), aux.stride(0), aux_mode, _stream())
    _lib.check(rc, "mil_gemm_aux")
    return out


def linear_bwd_params(dy, y, act: int, x, dW_out=None, db_out=None, want_db: bool = True, rows_dev=None):
    """dW = (dy (.) act'(y))^T x and db = its column sums in one product launch + fold (mil_linear_bwd_params).
    y may be None for act 0.  Returns (dW [N, K], db [N] or None)."""
    rows, N = dy.shape
    K = x.shape[1]
    dW = dW_out if dW_out is not None else torch.empty((N, K), device=dy.device, dtype=torch.float32)
    db = None
    if want_db:
        db = db_out if db_out is not None else torch.empty(N, device=dy.device, dtype=torch.float32)
    nws = _lib.lib().mil_linear_bwd_params_workspace_floats(rows, N, K)
    ws = torch.empty(nws, device=dy.device, dtype=torch.float32)
    rc = _lib.lib().mil_linear_bwd_params_rows(_p(dy), dy.stride(0), _p(y), y.stride(0) if y is not None else 0, act, _p(x),
                                               x.stride(0), rows, N, K, _p(dW), dW.stride(0), _p(db), 0, _p(ws), nws,
                                               _p(rows_dev), _stream())
    _lib.check(rc, "mil_linear_bwd_params_rows")
    return dW, db


def colsum(Y, out=None, accumulate: bool = False):
    M, N = Y.shape
    if out is None:
        out = torch.empty(N, device=Y.device, dtype=torch.float32)
    nws = _lib.lib().mil_colsum_workspace_floats(M, N)
    ws = torch.empty(nws, device=Y.device, dtype=torch.float32) if nws else None
    rc = _lib.lib().mil_colsum(_p(Y), Y.stride(0), M, N, _p(out), 1 if accumulate else 0, _p(ws), _stream())
    _lib.check(rc, "mil_colsum")
    return out


def act_bwd(dy, y, act: int):
    if act == 0:
        return dy
    dpre = torch.empty_like(dy)
    rc = _lib.lib().mil_act_bwd(_p(dy), _p(y), _p(dpre), dy.numel(), act, _stream())
    _lib.check(rc, "mil_act_bwd")
    return dpre


SMALL_ROWS = 64        # include/mil_hip.h: MIL_SMALL_ROWS (raising it to 512 was measured: T = 10 step 6.0 -> 6.7 ms)


def _small_ok(M: int, N: int, K: int, *tensors) -> bool:
    return (0 < M <= SMALL_ROWS and K % 16 == 0 and N % 16 == 0
            and all(t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0) for t in tensors))


MID_ROWS = 1024            # csrc/mid_linear.hip: one-launch products for layers between the few-rows and the tiled regime
MID_WORK = 340_000_000     # M * N * K up to which they beat the tiled GEMM + split-K fold (tools/kbench_mid.py)


def _mid_ok(M: int, N: int, K: int, *tensors) -> bool:
    return (SMALL_ROWS < M <= MID_ROWS and M * N * K <= MID_WORK and K % 8 == 0 and N % 8 == 0
            and all(t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0) for t in tensors))


def linear_mid_fwd(x, W, b, act: int, residual=None):
    """nn.Linear on 65..1024 rows in one launch (include/mil_hip.h: mil_linear_mid_fwd)."""
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mil_linear_mid_fwd(_p(x), x.stride(0), _p(W), W.stride(0), _p(b), act, _p(residual),
                                       residual.stride(0) if residual is not None else 0, _p(y), y.stride(0), M, N, K,
                                       _stream())
    _lib.check(rc, "mil_linear_mid_fwd")
    return y


def linear_mid_bwd(dy, y, act: int, x, W, need_dx: bool, need_dW: bool, need_db: bool, dW_out=None, db_out=None):
    """Backward of the same layer, one launch per product (mil_linear_mid_bwd): dx, dW (+ db from the same pass)."""
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    dx = torch.empty((M, K), device=dev, dtype=torch.float32) if need_dx else None
    need_dW = need_dW or need_db
    dW = (dW_out if dW_out is not None else torch.empty((N, K), device=dev, dtype=torch.float32)) if need_dW else None
    db = (db_out if db_out is not None else torch.empty(N, device=dev, dtype=torch.float32)) if need_db else None
    rc = _lib.lib().mil_linear_mid_bwd(_p(dy), dy.stride(0), _p(y) if act else None, y.stride(0) if act else 0, act, _p(x),
                                       x.stride(0), _p(W), W.stride(0), _p(dx), K, _p(dW), dW.stride(0) if need_dW else 0,
                                       _p(db), M, N, K, _stream())
    _lib.check(rc, "mil_linear_mid_bwd")
    return dx, dW, db


def _stream_int() -> int:
    return torch.cuda.current_stream().cuda_stream


def linear_small_fwd(x, W, b, act: int, residual=None, x2=None):
    """Token-side nn.Linear (M <= 64 rows) in one launch (include/mil_hip.h: mil_linear_small_fwd).  x2: a second addend of
    the input (mil_linear_small_fwd_add) - returns (y, x + x2)."""
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    if x2 is not None:
        xin = torch.empty((M, K), device=x.device, dtype=torch.float32)
        rc = _lib.lib().mil_linear_small_fwd_add(_p(x), x.stride(0), _p(x2), x2.stride(0), _p(xin), _p(W), W.stride(0), _p(b), act,
                                                 _p(residual), residual.stride(0) if residual is not None else 0, _p(y),
                                                 y.stride(0), M, N, K, _stream())
        _lib.check(rc, "mil_linear_small_fwd_add")
        return y, xin
    sh = _lib.shim()
    if sh is not None:          # same C entry through the torch cpp_extension binding (csrc/torch_shim.cpp)
        sh.linear_small_fwd(x, W, b, act, residual, y, _stream_int())
        return y
    rc = _lib.lib().mil_linear_small_fwd(_p(x), x.stride(0), _p(W), W.stride(0), _p(b), act, _p(residual),
                                         residual.stride(0) if residual is not None else 0, _p(y), y.stride(0),
                                         M, N, K, _stream())
    _lib.check(rc, "mil_linear_small_fwd")
    return y


def grad_slot(param):
    """The flat-gradient view optim.FlatAdam reserved for this parameter (None without FlatAdam): a backward that
    writes its result there and returns it hands autograd the final storage, so no gather copy is needed."""
    slot = getattr(param, "_mil_grad", None)
    if slot is None or slot.shape != param.shape:
        return None
    # Only the FIRST backward node of a pass may write the slot in place: were the same parameter used by a second node
    # (a shared Linear), its kernel would overwrite the first result before autograd's AccumulateGrad adds the two.  The
    # flag is cleared by FlatAdam.zero_grad() / GraphedStep; a second request in the same pass gets None, i.e. a fresh
    # tensor that autograd then accumulates into the slot.
    if getattr(param, "_mil_slot_used", False):
        # The first use's gradient may still be QUEUED (deferred.py forms it at the end of the pass): autograd is about to add
        # this node's fresh tensor to the slot, so whatever is queued for the slot has to be in it first (a Linear
        # used twice per forward - TwoWayTransformer_Both in the CT + pathology branch, aggregator.py:160-168 - lost the
        # first use's gradient and picked up stale slot contents instead).
        deferred.flush_pending()
        return None
    param._mil_slot_used = True
    # a fresh alias: autograd adopts a returned gradient without copying only if nothing else references that tensor
    return slot.detach()


def linear_small_bwd(dy, y_or_pre, act: int, x, W, want_dx: bool, want_dW: bool, want_db: bool, dW_out=None, db_out=None,
                     extras=(), dysum=None):
    """dx, dW, db of that layer in one launch (mil_linear_small_bwd).  extras: up to three more addends of dy (the gradients
    other consumers of the layer's output sent: _FanOut), summed while the operand is staged; dysum [M, N]: receives the sum."""
    M, K = x.shape
    N = W.shape[0]
    dx = torch.empty((M, K), device=x.device, dtype=torch.float32) if want_dx else None
    dW = (dW_out if dW_out is not None else torch.empty((N, K), device=x.device, dtype=torch.float32)) if want_dW else None
    db = (db_out if db_out is not None else torch.empty(N, device=x.device, dtype=torch.float32)) if want_db else None
    yv = y_or_pre if act != 0 else None
    if extras or dysum is not None:
        e = list(extras) + [None] * (3 - len(extras))
        rc = _lib.lib().mil_linear_small_bwd_sum(_p(dy), dy.stride(0), _p(e[0]), _p(e[1]), _p(e[2]), _p(dysum), _p(yv),
                                                 yv.stride(0) if yv is not None else 0, act, _p(x), x.stride(0), _p(W),
                                                 W.stride(0), _p(dx), K, _p(dW), K, _p(db), M, N, K, _stream())
        _lib.check(rc, "mil_linear_small_bwd_sum")
        return dx, dW, db
    sh = _lib.shim()
    if sh is not None:
        sh.linear_small_bwd(dy, yv, act, x, W, dx, dW, db, _stream_int())
        return dx, dW, db
    rc = _lib.lib().mil_linear_small_bwd(_p(dy), dy.stride(0), _p(yv), yv.stride(0) if yv is not None else 0, act,
                                         _p(x), x.stride(0), _p(W), W.stride(0), _p(dx), K, _p(dW), K, _p(db),
                                         M, N, K, _stream())
    _lib.check(rc, "mil_linear_small_bwd")
    return dx, dW, db


class _GradBox:
    """Mailbox between the node that PRODUCES a token-side tensor and the _FanOut node behind it: when the tensor has several
    consumers, _FanOut.backward leaves all but one of their gradients here and the producer's backward kernel sums them while
    it stages its operand (mil_linear_small_bwd_sum, mil_linear_small_ln_bwd3) - autograd would launch an elementwise add
    per extra consumer (6 of the fusion step's 93 launches)."""
    __slots__ = ("extras",)

    def __init__(self):
        self.extras = []

    def take(self, like=None):
        ex, self.extras = self.extras, []
        return [e for e in ex if e is not None]


def _sum_overflow(dy, extras, room: int):
    """extras beyond what the kernel takes are added the plain way; returns (dy, extras that fit)."""
    while len(extras) > room:
        dy = dy + extras.pop()
    return dy, extras


_ONES = {}


def backward(loss):
    """loss.backward() without autograd's fill launch for the root gradient: `ones_like(loss)` is a 4.7 us launch per step
    (inside every captured step); a cached constant per (device, shape) takes its place."""
    key = (loss.device, tuple(loss.shape), loss.dtype)
    one = _ONES.get(key)
    if one is None:
        if torch.cuda.is_current_stream_capturing():
            return loss.backward()          # no allocation + fill of a persistent constant inside a capture: the plain way
        one = _ONES[key] = torch.ones(loss.shape, device=loss.device, dtype=loss.dtype)
    return loss.backward(one)


def sum_n(ts):
    """Sum of 2 .. 4 same-shape fp32 tensors in ONE launch (mil_sum4); more, or odd layouts: torch's adds."""
    ts = list(ts)
    a = ts[0]
    ok = (2 <= len(ts) <= 4 and a.numel() % 4 == 0 and
          all(t.dtype == torch.float32 and t.is_contiguous() and t.shape == a.shape and t.data_ptr() % 16 == 0 for t in ts))
    if not ok:
        out = ts[0]
        for t in ts[1:]:
            out = out + t
        return out
    out = torch.empty_like(a)
    e = ts + [None] * (4 - len(ts))
    rc = _lib.lib().mil_sum4(_p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]), _p(out), a.numel(), _stream())
    _lib.check(rc, "mil_sum4")
    return out


def _ok_extra(e, like) -> bool:
    return e.dtype == torch.float32 and e.is_contiguous() and e.shape == like.shape and e.data_ptr() % 16 == 0


class _FanOut(torch.autograd.Function):
    """n aliases of x, one per consumer.  Backward: the first gradient goes up the graph as x's gradient, the others into the
    producer's mailbox (see _GradBox) - whatever autograd itself adds to the first one stays correct, sums commute."""

    @staticmethod
    def forward(ctx, x, box, n: int):
        ctx.box = box
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None, None
        first = gs[0]
        for g in gs[1:]:
            g = g if g.dtype == torch.float32 else g.float()
            ctx.box.extras.append(g.contiguous())
        return first, None, None


def fan_out(x, n: int):
    """n handles of x for n consumers.  When x came out of a token-side node that sums its output's gradients in-kernel
    (linear_act's few-rows path, lin_ln_lin's xn), the consumers' gradients meet there instead of in autograd's add launches;
    otherwise the handles are x itself."""
    box = getattr(x, "_mil_box", None)
    if n <= 1 or box is None or not torch.is_grad_enabled() or not x.requires_grad:
        return (x,) * n
    return _FanOut.apply(x, box, n)


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual): nn.Linear (+Tanh/ReLU) of aggregator.py:44-68, sam/transformer.py:413-416,
    sam/common.py:21-26.  x [M, K], W [N, K]."""

    @staticmethod
    def forward(ctx, x, W, b, act: int, residual, rows_dev=None, box=None, x2=None):
        x = _f32c(x, "x")
        W = _f32c(W, "W")
        M, K = x.shape
        ctx.box = box
        ctx.has_x2 = x2 is not None
        ctx.rows_dev = rows_dev             # capacity bucket: rows from rows_dev[0] on are padding (zero out, zero gradient)
        N = W.shape[0]
        res = _f32c(residual, "residual") if residual is not None else None
        pre = None
        ctx.small = _small_ok(M, N, K, x, W)
        ctx.mid = (not ctx.small) and _mid_ok(M, N, K, x, W, res) and \
            not (act == ACT["quickgelu"] and any(ctx.needs_input_grad[:3]))
        if residual is not None and act != 0:
            raise _lib.MilHipError("linear_act: residual is only supported with act='none'")
        if ctx.mid:
            y = linear_mid_fwd(x, W, b, act, res)
        elif ctx.small and not (act == ACT["quickgelu"] and any(ctx.needs_input_grad[:3])):
            if x2 is not None:
                y, x = linear_small_fwd(x, W, b, act, res, _f32c(x2, "x2"))      # x: the summed input, saved for the backward
            else:
                y = linear_small_fwd(x, W, b, act, res)
        elif act == ACT["quickgelu"] and any(ctx.needs_input_grad[:3]):
            # QuickGELU's derivative needs the pre-activation: keep it (learnable-prompt path only; the frozen
            # forward uses the fused epilogue)
            pre = linear_small_fwd(x, W, b, 0) if ctx.small else gemm(x, 0, W, 0, M, N, K, bias=b, act=0)
            y = torch.empty_like(pre)
            rc = _lib.lib().mil_quickgelu(_p(pre), None, _p(y), pre.numel(), _stream())
            _lib.check(rc, "mil_quickgelu")
        else:
            y = gemm(x, 0, W, 0, M, N, K, bias=b, act=act, residual=res, rows_dev=rows_dev)
        ctx.act = act
        ctx.has_b = b is not None
        ctx.b_param = b                     # only to look up its flat-gradient slot in backward
        ctx.has_res = residual is not None
        # with a residual the saved y is not the activation output; only act == none is used with residuals
        if x2 is not None and not (ctx.small and not ctx.mid):
            raise _lib.MilHipError("linear_act: x2 is only built for the few-rows path")
        ctx.save_for_backward(x, W, y if pre is None else pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        M, K = x.shape
        N = W.shape[0]
        W_slot = grad_slot(W)
        b_slot = grad_slot(ctx.b_param) if ctx.b_param is not None else None
        extras = ctx.box.take() if ctx.box is not None else []
        if extras and not (ctx.small and not ctx.mid and dy.is_contiguous() and all(_ok_extra(e, dy) for e in extras)):
            for e in extras:                    # another path, or an odd layout: the plain sums
                dy = dy + e
            extras = []
        if ctx.mid and ctx.act != ACT["quickgelu"]:
            if dy.data_ptr() % 16 or dy.stride(0) % 4:
                dy = dy.clone()
            dx, dW, db = linear_mid_bwd(dy, y, ctx.act, x, W, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                        ctx.has_b and ctx.needs_input_grad[2], W_slot, b_slot)
            return dx, (dW if ctx.needs_input_grad[1] else None), db, None, (dy if ctx.has_res else None), None, None, None
        if ctx.small:
            if dy.data_ptr() % 16:
                dy = dy.clone()
            dy, extras = _sum_overflow(dy, extras, 3)
            want_dx = ctx.needs_input_grad[0] or (ctx.has_x2 and ctx.needs_input_grad[7])
            if extras and not want_dx:
                # no dx launch to ride on (the layer's input carries no gradient): one n-ary sum, then the deferred dW
                dy, extras = sum_n([dy] + extras), []
            want_dW, want_db = ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2]
            if (deferred.enabled() and not want_dx and want_dW and W_slot is not None and (not want_db or b_slot is not None)
                    and x.shape[1] % 4 == 0 and dy.data_ptr() % 16 == 0):
                # nothing upstream waits for this layer: its whole backward is the grouped weight-gradient launch
                deferred.queue_dw(dy, y, x, W_slot.detach(), (b_slot.detach() if want_db else None), ctx.act)
                return None, W_slot, (b_slot if want_db else None), None, (dy if ctx.has_res else None), None, None, None
            if (deferred.enabled() and want_dx and want_dW and W_slot is not None and
                    (not want_db or b_slot is not None) and x.shape[1] % 4 == 0):
                # the previous layer's backward waits for dx only: dx now, the weight / bias gradient with every other
                # queued layer's in one grouped launch at the end of the pass (deferred.py), straight into the flat buffer.
                # The queue holds aliases of its own: autograd adopts a returned gradient without a copy only while
                # nothing else references that tensor object (see grad_slot)
                dysum = torch.empty((M, N), device=dy.device, dtype=torch.float32) if extras else None
                dx, _, _ = linear_small_bwd(dy, y, ctx.act, x, W, True, False, False, extras=extras, dysum=dysum)
                dyt = dysum if extras else dy
                deferred.queue_dw(dyt, y, x, W_slot.detach(), (b_slot.detach() if want_db else None), ctx.act)
                return (dx if ctx.needs_input_grad[0] else None, W_slot, (b_slot if want_db else None), None,
                        (dyt if ctx.has_res else None), None, None, (dx if ctx.has_x2 else None))
            if extras and ctx.has_res:
                for e in extras:                # the residual branch wants the summed gradient as a tensor
                    dy = dy + e
                extras = []
            dx, dW, db = linear_small_bwd(dy, y, ctx.act, x, W, want_dx, want_dW, want_db, W_slot, b_slot, extras=extras)
            return (dx if ctx.needs_input_grad[0] else None, dW, db, None, (dy if ctx.has_res else None), None, None,
                    (dx if ctx.has_x2 else None))
        if ctx.act == ACT["quickgelu"]:
            dpre = torch.empty_like(dy)
            rc = _lib.lib().mil_quickgelu(_p(y), _p(dy), _p(dpre), dy.numel(), _stream())     # y holds the pre-activation
            _lib.check(rc, "mil_quickgelu")
        else:
            fused = ctx.needs_input_grad[1] and N % 4 == 0 and K % 4 == 0
            if fused and not ctx.needs_input_grad[0]:
                # parameters only (fc_pathology: the bag features carry no gradient): act' is applied while dy is staged
                # for the weight-gradient product, which also yields the bias gradient - no dpre tensor at all
                dW, db = linear_bwd_params(dy, y if ctx.act else None, ctx.act, x, W_slot, b_slot,
                                           ctx.has_b and ctx.needs_input_grad[2], rows_dev=ctx.rows_dev)
                return None, dW, db, None, (dy if ctx.has_res else None), None, None, None
            dpre = act_bwd(dy, y, ctx.act)
            if fused:
                dx = gemm(dpre, 0, W, 1, M, K, N)
                dW, db = linear_bwd_params(dpre, None, 0, x, W_slot, b_slot, ctx.has_b and ctx.needs_input_grad[2])
                return dx, dW, db, None, (dy if ctx.has_res else None), None, None, None
        dx = gemm(dpre, 0, W, 1, M, K, N) if ctx.needs_input_grad[0] else None
        dW = gemm(dpre, 1, x, 1, N, K, M, out=W_slot, split_k=True) if ctx.needs_input_grad[1] else None
        db = colsum(dpre, out=b_slot) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        dres = dy if ctx.has_res else None
        return dx, dW, db, None, dres, None, None, None


def _small_dw(dy, yv, x, W, b, act: int):
    """(dW, db) of a few-rows layer: queued for the end-of-backward grouped launch straight into the flat gradient slots
    when optim.FlatAdam owns the parameters (deferred.py), otherwise formed now (mil_linear_small_bwd, dW role only)."""
    W_slot = grad_slot(W)
    b_slot = grad_slot(b) if b is not None else None
    if (deferred.enabled() and W_slot is not None and (b is None or b_slot is not None) and x.shape[1] % 4 == 0
            and dy.data_ptr() % 16 == 0):
        deferred.queue_dw(dy, yv, x, W_slot.detach(), (b_slot.detach() if b is not None else None), act)
        return W_slot, b_slot
    _, dW, db = linear_small_bwd(dy, yv, act, x, W, False, True, b is not None, W_slot, b_slot)
    return dW, db


class _LinLnLin(torch.autograd.Function):
    """u = z Wp^T + bp (+ resp);  xn = LayerNorm(u);  y = act((xn [+ x2]) Wc^T + bc)  - the P -> norm -> C links of the
    two-way block's token stream (sam/transformer.py:287-300) as ONE autograd node of two launches forward
    (mil_linear_small_fwd, mil_linear_small_ln_fwd) and two backward (C's input gradient; P's input gradient with the norm's
    backward and the sum of the two gradients that reach xn applied while its operand is staged, mil_linear_small_ln_bwd) -
    the op-by-op route takes three and five (a LayerNorm launch each way and autograd's add).  Returns (y, xn): later
    consumers of the norm's output use xn, whose gradient arrives here.  Weight gradients join the grouped launch."""

    @staticmethod
    def forward(ctx, z, Wp, bp, resp, gamma, beta, eps: float, x2, Wc, bc, actc: int, box=None):
        ctx.box = box
        z, Wp, Wc = _f32c(z, "z"), _f32c(Wp, "Wp"), _f32c(Wc, "Wc")
        M = z.shape[0]
        dev = z.device
        resp_c = _f32c(resp, "residual") if resp is not None else None
        u = linear_small_fwd(z, Wp, bp, 0, resp_c)
        E, N = u.shape[1], Wc.shape[0]
        y = torch.empty((M, N), device=dev, dtype=torch.float32)
        xn = torch.empty((M, E), device=dev, dtype=torch.float32)
        x2c = _f32c(x2, "x2") if x2 is not None else None
        xin = torch.empty((M, E), device=dev, dtype=torch.float32) if x2c is not None else None
        stats = torch.empty((M, 2), device=dev, dtype=torch.float32)
        rc = _lib.lib().mil_linear_small_ln_fwd(_p(u), u.stride(0), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")), float(eps),
                                                _p(x2c), x2c.stride(0) if x2c is not None else 0, _p(Wc), Wc.stride(0), _p(bc),
                                                int(actc), None, 0, _p(y), N, _p(xn), _p(xin), _p(stats), M, N, _stream())
        _lib.check(rc, "mil_linear_small_ln_fwd")
        ctx.actc, ctx.has_res, ctx.has_x2 = int(actc), resp is not None, x2 is not None
        ctx.params = (bp, beta, bc)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, Wp, u, stats, gamma, xin if xin is not None else xn, Wc, y)
        return y, xn

    @staticmethod
    def backward(ctx, dy, dxn):
        z, Wp, u, stats, gamma, xin, Wc, y = ctx.saved_tensors
        bp, beta, bc = ctx.params
        M, E = u.shape
        dev = u.device
        if dy is not None:
            dy = _f32c(dy, "dy")
            if dy.data_ptr() % 16:
                dy = dy.clone()
            parts = None
            Nc = Wc.shape[0]
            if Nc >= 2048 and Nc % 2048 == 0 and not ctx.has_x2 and dy.stride(0) % 4 == 0 and y.stride(0) % 4 == 0:
                # a wide C layer (mlp.lin1): its input gradient as four partial sums over n (4 x the workgroups, one operand
                # chunk each); the norm's backward below adds them while it stages its operand
                parts = torch.empty((4, M, E), device=dev, dtype=torch.float32)
                rc = _lib.lib().mil_linear_small_bwd_split(_p(dy), dy.stride(0), _p(y if ctx.actc != 0 else None), y.stride(0),
                                                           ctx.actc, _p(Wc), Wc.stride(0), _p(parts), M, Nc, E, 4, _stream())
                _lib.check(rc, "mil_linear_small_bwd_split")
                dxin = None
            else:
                dxin, _, _ = linear_small_bwd(dy, y, ctx.actc, xin, Wc, True, False, False)  # C: input gradient now ...
            dWc, dbc = _small_dw(dy, y, xin, Wc, bc, ctx.actc)                              # ... weight gradient grouped
        else:
            dxin, dWc, dbc, parts = None, None, None, None
        gs = [g for g in (dxin, dxn) if g is not None] + (ctx.box.take() if ctx.box is not None else [])
        if parts is not None:
            gs = [parts[0], parts[1], parts[2]] + gs + [parts[3]]        # slots 4 and 5 take contiguous [M, 512] addends
        if not gs:
            return (None,) * 12
        gs = [_f32c(g, "dxn") for g in gs]
        def fits(i, g):           # addends 1-3 carry their own row stride, 4-5 are read as contiguous [M, 512]
            return g.stride(1) == 1 and g.data_ptr() % 16 == 0 and (g.stride(0) % 4 == 0 if i < 3 else g.is_contiguous())
        i = 1
        while i < len(gs):        # whatever does not fit a slot is added the plain way
            if i >= 5 or not fits(i, gs[i]):
                gs[0] = gs[0] + gs.pop(i)
            else:
                i += 1
        if not fits(0, gs[0]):
            gs[0] = gs[0].contiguous()
        g1, g2, g3, g4, g5 = (gs + [None] * 4)[:5]
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        du = torch.empty((M, E), device=dev, dtype=torch.float32)
        dg = grad_slot(gamma)
        if dg is None:
            dg = torch.empty(E, device=dev, dtype=torch.float32)
        db = grad_slot(beta)
        if db is None:
            db = torch.empty(E, device=dev, dtype=torch.float32)
        K = z.shape[1]
        rc = _lib.lib().mil_linear_small_ln_bwd5(_p(g1), g1.stride(0), _p(g2), g2.stride(0) if g2 is not None else 0, _p(g3),
                                                 g3.stride(0) if g3 is not None else 0, _p(g4), _p(g5), _p(u), u.stride(0),
                                                 _p(stats), _p(gamma), _p(Wp), Wp.stride(0), _p(dz), K, _p(du), _p(dg), _p(db),
                                                 M, K, _stream())
        _lib.check(rc, "mil_linear_small_ln_bwd5")
        dWp, dbp = _small_dw(du, None, z, Wp, bp, 0)
        return (dz, dWp, dbp, (du if ctx.has_res else None), dg, db, None, (dxin if ctx.has_x2 else None), dWc, dbc, None, None)


def lin_ln_lin_ok(z, Wp, gamma, Wc) -> bool:
    """Shapes the fused P -> LayerNorm -> C node is built for: <= 64 rows, norm width 512, 16-byte aligned operands."""
    return (z.dim() == 2 and 0 < z.shape[0] <= SMALL_ROWS and Wp.shape[0] == 512 and gamma.shape[0] == 512 and
            Wc.shape[1] == 512 and Wc.shape[0] % 16 == 0 and z.shape[1] % 16 == 0 and
            _small_ok(z.shape[0], Wp.shape[0], z.shape[1], z, Wp) and Wc.data_ptr() % 16 == 0 and
            gamma.requires_grad and Wp.requires_grad and Wc.requires_grad)


def lin_ln_lin(z, Wp, bp, resp, gamma, beta, eps, x2, Wc, bc, actc: str = "none"):
    """(y, xn) of _LinLnLin; see there.  xn carries the node's mailbox: hand it to several consumers through fan_out()."""
    box = _GradBox()
    y, xn = _LinLnLin.apply(z, Wp, bp, resp, gamma, beta, float(eps), x2, Wc, bc, ACT[actc], box)
    xn._mil_box = box
    return y, xn


class _MlpQuickGelu(torch.autograd.Function):
    """x + c_proj(QuickGELU(c_fc(x_ln)))  (clip/model.py:176-178,196-198) as one autograd node: the c_fc product stores
    its pre-activation from the epilogue, and in the backward the product dout . W2 is multiplied by QuickGELU' in its
    epilogue - no stand-alone activation forward / backward passes over the [rows, 4 W] tensors."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, residual):
        x, W1, W2 = _f32c(x, "x"), _f32c(W1, "W1"), _f32c(W2, "W2")
        M, K = x.shape
        N1 = W1.shape[0]
        need = any(ctx.needs_input_grad[:5])
        if need:
            pre = torch.empty((M, N1), device=x.device, dtype=torch.float32)
            h = gemm_aux(x, W1, 0, M, N1, K, pre, 1, bias=b1, act=ACT["quickgelu"])
        else:
            pre = None
            h = gemm(x, 0, W1, 0, M, N1, K, bias=b1, act=ACT["quickgelu"])
        out = gemm(h, 0, W2, 0, M, W2.shape[0], N1, bias=b2, residual=_f32c(residual, "residual") if residual is not None else None)
        ctx.has_res = residual is not None
        keep_h = ctx.needs_input_grad[3]
        ctx.save_for_backward(x, W1, W2, pre, h if keep_h else None)
        ctx.params = (b1, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W1, W2, pre, h = ctx.saved_tensors
        b1, b2 = ctx.params
        dout = _f32c(dout, "dout")
        M, K = x.shape
        N1, N2 = W1.shape[0], W2.shape[0]
        dpre = gemm_aux(dout, W2, 1, M, N1, N2, pre, 2)                      # (dout . W2) * QuickGELU'(pre)
        dx = gemm(dpre, 0, W1, 1, M, K, N1) if ctx.needs_input_grad[0] else None
        dW1 = gemm(dpre, 1, x, 1, N1, K, M, out=grad_slot(W1)) if ctx.needs_input_grad[1] else None
        db1 = colsum(dpre, out=grad_slot(b1)) if (b1 is not None and ctx.needs_input_grad[2]) else None
        dW2 = gemm(dout, 1, h, 1, N2, N1, M, out=grad_slot(W2)) if ctx.needs_input_grad[3] else None
        db2 = colsum(dout, out=grad_slot(b2)) if (b2 is not None and ctx.needs_input_grad[4]) else None
        return dx, dW1, db1, dW2, db2, (dout if ctx.has_res else None)


def mlp_quickgelu(x, W1, b1, W2, b2, residual=None):
    """c_proj(QuickGELU(c_fc(x))) + residual; tall inputs take the fused node, a few rows the one-launch kernels."""
    if x.shape[0] <= SMALL_ROWS:
        return linear_act(linear_act(x, W1, b1, "quickgelu"), W2, b2, "none", residual=residual)
    return _MlpQuickGelu.apply(x, W1, b1, W2, b2, residual)


def linear_act(x, W, b=None, act: str = "none", residual=None, rows_dev=None, x2=None):
    """rows_dev: device int32 [1] with the true row count of a capacity bucket (x has the bucket's capacity rows; the rows behind
    the count come out as zeros and carry no gradient) - tall layers only; the few-rows kernels ignore it.
    x2: the layer's input is x + x2 (few-rows layers add it while the operand is staged; others through a plain add).
    The result carries the node's gradient mailbox (see fan_out)."""
    lead = x.shape[:-1]
    x2d = x.reshape(-1, x.shape[-1])
    if x2 is not None:
        x2 = x2.reshape(-1, x2.shape[-1])
        M, K = x2d.shape
        N = W.shape[0]
        few = (x2d.is_cuda and x2d.dtype == torch.float32 and x2.dtype == torch.float32 and x2d.is_contiguous() and x2.is_contiguous()
               and W.dtype == torch.float32 and W.is_contiguous()
               and _small_ok(M, N, K, x2d, W) and not _mid_ok(M, N, K, x2d, W, residual) and act != "quickgelu")
        if not few:
            x2d, x2 = x2d + x2, None
    box = _GradBox()
    y = _LinearAct.apply(x2d, W, b, ACT[act], residual.reshape(-1, residual.shape[-1]) if residual is not None else None, rows_dev,
                         box, x2)
    y = y.reshape(*lead, W.shape[0])
    y._mil_box = box
    return y


# --------------------------------------------------------------------------- K2: attention cores, LayerNorm, PE
def _head_dim(I: int, H: int) -> int:
    c = I // H
    if c * H != I or c not in (32, 64):
        raise _lib.MilHipError(f"attention: internal dim {I} / {H} heads = head dim {c}; kernels support 32 and 64")
    return c


SEQ_MAX_TOKENS = 96      # csrc/attention.hip: AS_MAXT


class _AttnRows(torch.autograd.Function):
    """softmax(q k^T / sqrt(c)) v, one thread per (query row, head): sam/transformer.py:441-446 for
    image->token / token self attention, clip/model.py:183 with causal=True."""

    @staticmethod
    def forward(ctx, q, k, v, segs, H: int, causal: bool):
        q, k, v = _f32c(q, "q"), _f32c(k, "k"), _f32c(v, "v")
        Tq, I = q.shape
        C = _head_dim(I, H)
        o = torch.empty_like(q)
        lse = torch.empty((Tq, H), device=q.device, dtype=torch.float32)
        if 16 < segs.Tk_max <= SEQ_MAX_TOKENS and segs.q_lengths == segs.k_lengths:
            # whole-sequence self-attention (CLIP text blocks): LDS-staged heads on MFMA, one workgroup per (sequence, head)
            rc = _lib.lib().mil_attn_seq_fwd(_p(q), _p(k), _p(v), I, _p(segs.q_off), segs.B, segs.Tq_max, H, C,
                                             1 if causal else 0, _p(o), _p(lse), _stream())
            _lib.check(rc, "mil_attn_seq_fwd")
        else:
            rc = _lib.lib().mil_attn_rows_fwd(_p(q), _p(k), _p(v), _p(segs.q_off), _p(segs.k_off), _p(segs.q_bag), Tq, H, C,
                                              1 if causal else 0, _p(o), _p(lse), _stream())
            _lib.check(rc, "mil_attn_rows_fwd")
        ctx.segs, ctx.H, ctx.C, ctx.causal = segs, H, C, causal
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        segs, H, C = ctx.segs, ctx.H, ctx.C
        I = H * C
        do = _f32c(do, "do")
        if (not ctx.causal) and segs.Tk_max > 16 and (segs.q_lengths != segs.k_lengths or segs.Tk_max > SEQ_MAX_TOKENS):
            # more than 16 keys per bag outside whole-sequence self-attention (`--alignment_base CT`: 160 CT tokens as
            # queries / keys): the general per-(row, head) loops (include/mil_hip.h: mil_attn_rows_bwd_general)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            ws = torch.empty(max(1, q.shape[0]) * H, device=q.device, dtype=torch.float32)
            rc = _lib.lib().mil_attn_rows_bwd_general(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(segs.q_off), _p(segs.k_off),
                                                      _p(segs.q_bag), _p(segs.k_bag), q.shape[0], k.shape[0], H, C, _p(dq),
                                                      _p(dk), _p(dv), _p(ws), _stream())
            _lib.check(rc, "mil_attn_rows_bwd_general")
            return dq, dk, dv, None, None, None
        if ctx.causal or segs.Tk_max > 16:
            # whole-sequence self-attention (the CLIP text blocks under learnable prompts): q, k, v share the segments
            if segs.q_lengths != segs.k_lengths or segs.Tk_max > SEQ_MAX_TOKENS:
                raise _lib.MilHipError("attention backward: > 16 keys per bag is only supported for self-attention over "
                                       f"sequences of <= {SEQ_MAX_TOKENS} tokens")
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            rc = _lib.lib().mil_attn_seq_bwd(_p(q), _p(k), _p(v), I, _p(o), _p(do), _p(lse), _p(segs.q_off), segs.B,
                                             segs.Tq_max, H, C, 1 if ctx.causal else 0, _p(dq), _p(dk), _p(dv), I,
                                             _stream())
            _lib.check(rc, "mil_attn_seq_bwd")
            return dq, dk, dv, None, None, None
        dq, dk, dv = torch.empty_like(q), torch.zeros_like(k), torch.zeros_like(v)
        ws = torch.empty(max(1, segs.nblk) * 2 * 16 * I, device=q.device, dtype=torch.float32)
        rc = _lib.lib().mil_attn_rows_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(segs.k_off), _p(segs.blk_map),
                                          _p(segs.bag_blk_off), segs.nblk, segs.B, H, C, _p(dq), _p(dk), _p(dv), _p(ws),
                                          _stream())
        _lib.check(rc, "mil_attn_rows_bwd")
        return dq, dk, dv, None, None, None


def attention_rows(q, k, v, segs, H: int, causal: bool = False):
    return _AttnRows.apply(q, k, v, segs, H, causal)


class _AttnSeqPacked(torch.autograd.Function):
    """Whole-sequence self-attention on the PACKED in_proj output qkv [rows, 3 W] (clip/model.py:171-178: one
    nn.MultiheadAttention in_proj of width 3 W): the kernels read q / k / v as column blocks (row stride 3 W) and the
    backward writes dq | dk | dv into one [rows, 3 W] tensor, so the projection and its backward are ONE GEMM each."""

    @staticmethod
    def forward(ctx, qkv, segs, H: int, causal: bool):
        qkv = _f32c(qkv, "qkv")
        rows, W3 = qkv.shape
        W = W3 // 3
        C = _head_dim(W, H)
        o = torch.empty((rows, W), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((rows, H), device=qkv.device, dtype=torch.float32)
        base = qkv.data_ptr()
        rc = _lib.lib().mil_attn_seq_fwd(base, base + 4 * W, base + 8 * W, W3, _p(segs.q_off), segs.B, segs.Tq_max, H, C,
                                         1 if causal else 0, _p(o), _p(lse), _stream())
        _lib.check(rc, "mil_attn_seq_fwd")
        ctx.segs, ctx.H, ctx.C, ctx.causal = segs, H, C, causal
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        segs, H, C = ctx.segs, ctx.H, ctx.C
        W3 = qkv.shape[1]
        W = W3 // 3
        do = _f32c(do, "do")
        dqkv = torch.empty_like(qkv)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        rc = _lib.lib().mil_attn_seq_bwd(base, base + 4 * W, base + 8 * W, W3, _p(o), _p(do), _p(lse), _p(segs.q_off),
                                         segs.B, segs.Tq_max, H, C, 1 if ctx.causal else 0, dbase, dbase + 4 * W,
                                         dbase + 8 * W, W3, _stream())
        _lib.check(rc, "mil_attn_seq_bwd")
        return dqkv, None, None, None


def seq_attention_ok(segs) -> bool:
    return 16 < segs.Tk_max <= SEQ_MAX_TOKENS and segs.q_lengths == segs.k_lengths


def attention_seq_packed(qkv, segs, H: int, causal: bool = False):
    return _AttnSeqPacked.apply(qkv, segs, H, causal)


class _AttnPool(torch.autograd.Function):
    """<= 16 queries per bag over many keys (token->image attention, sam/transformer.py:293,116)."""

    @staticmethod
    def forward(ctx, q, k, v, segs, H: int):
        q, k, v = _f32c(q, "q"), _f32c(k, "k"), _f32c(v, "v")
        Tq, I = q.shape
        C = _head_dim(I, H)
        if segs.Tq_max > 16:
            raise _lib.MilHipError("attention pool form supports <= 16 queries per bag")
        o = torch.empty_like(q)
        lse = torch.empty((Tq, H), device=q.device, dtype=torch.float32)
        ws = torch.empty(max(1, segs.ntiles) * 16 * (I + 2 * H), device=q.device, dtype=torch.float32)
        rc = _lib.lib().mil_attn_pool_fwd_mh(_p(q), _p(k), _p(v), _p(segs.q_off), _p(segs.tile_map), _p(segs.bag_tile_off),
                                             segs.ntiles, segs.B, max(1, segs.Tq_max), H, C, _p(o), _p(lse), _p(ws), _stream())
        _lib.check(rc, "mil_attn_pool_fwd_mh")
        ctx.segs, ctx.H, ctx.C = segs, H, C
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        segs, H, C = ctx.segs, ctx.H, ctx.C
        I = H * C
        do = _f32c(do, "do")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(max(1, segs.ntiles) * 16 * I, device=q.device, dtype=torch.float32)
        rc = _lib.lib().mil_attn_pool_bwd_mh(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(segs.q_off), _p(segs.tile_map),
                                             _p(segs.bag_tile_off), segs.ntiles, segs.B, max(1, segs.Tq_max), H, C, _p(dq),
                                             _p(dk), _p(dv), _p(ws), _stream())
        _lib.check(rc, "mil_attn_pool_bwd_mh")
        return dq, dk, dv, None, None


def attention_pool(q, k, v, segs, H: int):
    if segs.Tq_max > 16:          # more queries per bag than the pool form holds: the rows form (general backward)
        return _AttnRows.apply(q, k, v, segs, H, False)
    return _AttnPool.apply(q, k, v, segs, H)


def _tail_view(base, rows: int):
    """The `rows` rows reserved BEHIND `base` in its storage (layer_norm(..., tail_rows=rows) allocated them), or None."""
    if base is None or base.dim() != 2 or not base.is_contiguous() or base.storage_offset() != 0 or base.dtype != torch.float32:
        return None
    R, E = base.shape
    st = base.untyped_storage()
    if st.nbytes() != (R + rows) * E * 4:
        return None
    return torch.empty(0, device=base.device, dtype=torch.float32).set_(st, R * E, (rows, E), (E, 1))


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, tail_rows: int = 0, into_tail_of=None):
        x = _f32c(x, "x")
        rows, E = x.shape
        # tail_rows > 0: allocate room for that many more rows behind the result (see append_rows)
        y = torch.empty((rows + tail_rows, E), device=x.device, dtype=torch.float32)[:rows] if tail_rows else None
        if y is None and into_tail_of is not None:
            # the result IS the block another tensor reserved behind itself (the text tokens of the multi-modal bag,
            # model/aggregator.py:192): written in place there, append_rows then has nothing to copy
            y = _tail_view(into_tail_of.detach(), rows) if into_tail_of.shape[1] == E else None
        if y is None:
            y = torch.empty_like(x)
        stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        sh = _lib.shim()
        if sh is not None:
            sh.layernorm_fwd(x, _f32c(gamma, "gamma"), _f32c(beta, "beta"), eps, y, stats, _stream_int())
        else:
            rc = _lib.lib().mil_layernorm_fwd(_p(x), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")), rows, E, eps, _p(y),
                                              _p(stats), _stream())
            _lib.check(rc, "mil_layernorm_fwd")
        ctx.save_for_backward(x, gamma, stats)
        ctx.beta_param = beta               # only to look up its flat-gradient slot in backward
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        dx, dg, db = _layer_norm_bwd(x, gamma, stats, dy, None, ctx.needs_input_grad[1] or ctx.needs_input_grad[2],
                                     ctx.beta_param)
        return dx, dg, db, None, None, None


def _layer_norm_bwd(x, gamma, stats, dy, dres, want_params: bool, beta=None):
    """dx (+ dres), dgamma, dbeta of a LayerNorm; frozen parameters skip their sums (and the two column-sum launches)."""
    rows, E = x.shape
    dy = _f32c(dy, "dy")
    dx = torch.empty_like(x)
    dg = db = ws = None
    if want_params:
        dg = grad_slot(gamma)
        if dg is None:
            dg = torch.empty(E, device=x.device, dtype=torch.float32)
        db = grad_slot(beta) if beta is not None else None
        if db is None:
            db = torch.empty(E, device=x.device, dtype=torch.float32)
        ws = torch.empty(_lib.lib().mil_layernorm_bwd_blocks(rows) * 2 * E, device=x.device, dtype=torch.float32)
    sh = _lib.shim()
    if sh is not None:
        sh.layernorm_bwd_res(x, gamma, dy, stats, dres, dx, dg, db, ws, _stream_int())
        return dx, dg, db
    rc = _lib.lib().mil_layernorm_bwd_res(_p(x), _p(gamma), _p(dy), _p(stats), _p(dres), rows, E, _p(dx), _p(dg), _p(db),
                                          _p(ws), _stream())
    _lib.check(rc, "mil_layernorm_bwd_res")
    return dx, dg, db


class _LayerNormRes(torch.autograd.Function):
    """LayerNorm that also hands back its input (an alias): a pre-norm residual block x + f(LN(x)) (clip/model.py:183-199)
    passes THAT to the residual add, so the gradient of the residual branch arrives at this node and is added inside the
    backward kernel instead of by an elementwise launch of autograd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        x_in = x
        x = _f32c(x, "x")
        rows, E = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        rc = _lib.lib().mil_layernorm_fwd(_p(x), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")), rows, E, eps, _p(y),
                                          _p(stats), _stream())
        _lib.check(rc, "mil_layernorm_fwd")
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, gamma, stats)
        return y, x_in.view_as(x_in)

    @staticmethod
    def backward(ctx, dy, dres):
        if dy is None:
            return dres, None, None, None
        x, gamma, stats = ctx.saved_tensors
        dres = _f32c(dres, "dres") if dres is not None else None
        dx, dg, db = _layer_norm_bwd(x, gamma, stats, dy, dres, ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dg, db, None


def layer_norm_res(x, gamma, beta, eps: float = 1e-5):
    """(LayerNorm(x), x) for a 2-D x - use the returned x for the residual add that skips the norm."""
    return _LayerNormRes.apply(x, gamma, beta, eps)


class _LayerNormBagRow(torch.autograd.Function):
    """LayerNorm(x + o[bag of row]) as one node (mil_layernorm_bagrow_fwd/_bwd): the sum is never materialised, and the
    backward returns dx, the per-bag sums d_o, dgamma and dbeta from one pass + one fold launch."""

    @staticmethod
    def forward(ctx, x, o, gamma, beta, eps: float, segs, tail_rows: int):
        x, o = _f32c(x, "x"), _f32c(o, "o")
        rows, E = x.shape
        y = torch.empty((rows + tail_rows, E), device=x.device, dtype=torch.float32)[:rows] if tail_rows else torch.empty_like(x)
        stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        rc = _lib.lib().mil_layernorm_bagrow_fwd(_p(x), _p(o), _p(segs.q_bag), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")),
                                                 rows, E, eps, _p(y), _p(stats), _stream())
        _lib.check(rc, "mil_layernorm_bagrow_fwd")
        ctx.segs = segs
        ctx.beta_param = beta
        ctx.save_for_backward(x, o, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, o, gamma, stats = ctx.saved_tensors
        segs = ctx.segs
        dy = _f32c(dy, "dy")
        rows, E = x.shape
        dx = torch.empty_like(x)
        do = torch.empty_like(o)
        dg = grad_slot(gamma)
        if dg is None:
            dg = torch.empty(E, device=x.device, dtype=torch.float32)
        db = grad_slot(ctx.beta_param)
        if db is None:
            db = torch.empty(E, device=x.device, dtype=torch.float32)
        ws = torch.empty(_lib.lib().mil_layernorm_bwd_blocks(rows) * 4 * E, device=x.device, dtype=torch.float32)
        rc = _lib.lib().mil_layernorm_bagrow_bwd(_p(x), _p(o), _p(segs.q_bag), _p(segs.q_off), segs.B, _p(gamma), _p(dy),
                                                 _p(stats), rows, E, _p(dx), _p(do), _p(dg), _p(db), _p(ws), _stream())
        _lib.check(rc, "mil_layernorm_bagrow_bwd")
        return dx, do, dg, db, None, None, None


def layer_norm_bag_row(x, o, segs, gamma, beta, eps: float = 1e-5, tail_rows: int = 0):
    """LayerNorm(x + o[bag of row]) for x [rows, E], o [B, E]; segs: AttnSegs whose QUERY side are the rows of x.  Fused
    when every bag is at least one backward workgroup's row range long (see mil_layernorm_bagrow_bwd) and the norm is
    trainable; otherwise add_bag_row followed by layer_norm."""
    rows = x.shape[0]
    # device-side lengths (segments.FusionBucket): set_lengths() has checked every bag against the block's row range
    ok = (x.dim() == 2 and rows > 64 and gamma.requires_grad == beta.requires_grad and
          (getattr(segs, "device_lengths", False) or
           min(segs.q_lengths, default=0) >= _lib.lib().mil_layernorm_bagrow_rows_per_block(rows)))
    if not ok:
        return layer_norm(add_bag_row(x, o, segs), gamma, beta, eps, tail_rows)
    return _LayerNormBagRow.apply(x, o, gamma, beta, eps, segs, tail_rows)


def layer_norm(x, gamma, beta, eps: float = 1e-5, tail_rows: int = 0, into_tail_of=None):
    lead = x.shape[:-1]
    return _LayerNorm.apply(x.reshape(-1, x.shape[-1]), gamma, beta, eps, tail_rows, into_tail_of).reshape(*lead, x.shape[-1])


class _AppendRows(torch.autograd.Function):
    """torch.cat([base, extra], 0) that does not copy `base` when it was allocated with room behind it
    (layer_norm(..., tail_rows=extra.shape[0])): only the few `extra` rows are written.  Used for the multi-modal
    bag of model/aggregator.py:192 - the [N, 512] patch tokens stay where the last LayerNorm put them."""

    @staticmethod
    def forward(ctx, base, extra, tail_reserved: bool):
        R, E = base.shape
        T = extra.shape[0]
        ctx.R = R
        st = base.untyped_storage()
        # tail_reserved is the CALLER's statement that the rows behind `base` were reserved for this purpose
        # (layer_norm(..., tail_rows=T)); without it a prefix view of somebody else's buffer would qualify too.
        if (tail_reserved and base.is_contiguous() and base.storage_offset() == 0 and base.dtype == torch.float32
                and st.nbytes() == (R + T) * E * 4):
            big = torch.empty(0, device=base.device, dtype=torch.float32).set_(st, 0, (R + T, E), (E, 1))
            if not (extra.is_contiguous() and extra.dtype == torch.float32 and extra.data_ptr() == base.data_ptr() + R * E * 4):
                big[R:].copy_(extra)          # (else: the producer already wrote them there - layer_norm(into_tail_of=base))
            return big
        return torch.cat([base, extra], 0)

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.R], g[ctx.R:], None


def append_rows(base, extra, tail_reserved: bool = False):
    """torch.cat([base, extra], 0); with tail_reserved=True and `base` allocated by layer_norm(..., tail_rows=len(extra))
    the base rows are not copied."""
    return _AppendRows.apply(base, extra, tail_reserved)


class _AddPE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe, row_bag, row_off):
        x = _f32c(x, "x")
        rows, E = x.shape
        out = torch.empty_like(x)
        rc = _lib.lib().mil_add_pe(_p(x), _p(pe), _p(row_bag), _p(row_off), rows, E, _p(out), _stream())
        _lib.check(rc, "mil_add_pe")
        ctx.set_materialize_grads(False)      # an unused keys + pe must not send a [rows, E] tensor of zeros upstream
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


def add_pe(x, pe, row_bag, row_off):
    """x[row] + pe[position of the row inside its bag]."""
    return _AddPE.apply(x, pe, row_bag, row_off)


def ct_map_tokens(ct, model_CT: str = "resnetMC3_18"):
    """CT feature map [B, C, D, h, w] -> flat token rows [B * T, C] (sam/transformer.py:86-98): T = D tokens by a mean over
    (h, w) for resnetMC3_18, T = D * h * w by flatten + permute for medicalNet.  The map is an input (the CT encoders are
    outside the hot path), so no gradient flows into it."""
    if ct.dim() != 5:
        raise _lib.MilHipError("ct_map_tokens: expected a 5-D map [B, C, D, h, w]")
    if ct.requires_grad:
        raise NotImplementedError("the CT encoder is outside the MIL hot path: pass its output detached")
    ct = _f32c(ct, "ct")
    B, C, D, hh, ww = ct.shape
    reduce = 0 if model_CT == "medicalNet" else 1
    T = D if reduce else D * hh * ww
    out = torch.empty((B * T, C), device=ct.device, dtype=torch.float32)
    rc = _lib.lib().mil_ct_map_tokens(_p(ct), B, C, D, hh * ww, reduce, _p(out), _stream())
    _lib.check(rc, "mil_ct_map_tokens")
    return out, T


def sinusoid_pe(n: int, E: int, device):
    pe = torch.empty((n, E), device=device, dtype=torch.float32)
    rc = _lib.lib().mil_sinusoid_pe(_p(pe), n, E, _stream())
    _lib.check(rc, "mil_sinusoid_pe")
    return pe


# --------------------------------------------------------------------------- K4: CLIP text front/back ends
def embed_tokens(ids, table, pos):
    """token_embedding[ids] + positional_embedding  ->  [nseq * ctx, W]  (clip/model.py:340-342)."""
    nseq, ctx = ids.shape
    W = table.shape[1]
    if ids.dtype != torch.int64 or not ids.is_cuda:
        raise _lib.MilHipError("embed_tokens: ids must be int64 on the GPU")
    out = torch.empty((nseq * ctx, W), device=ids.device, dtype=torch.float32)
    rc = _lib.lib().mil_embed_tokens(_p(ids.contiguous()), _p(_f32c(table, "table")), _p(_f32c(pos, "pos")), nseq, ctx, W,
                                     _p(out), _stream())
    _lib.check(rc, "mil_embed_tokens")
    return out


def gather_eot(ids, x):
    """Rows of x [nseq*ctx, W] at each sequence's EOT position (argmax of the ids)."""
    nseq, ctx = ids.shape
    W = x.shape[1]
    out = torch.empty((nseq, W), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mil_gather_eot(_p(ids.contiguous()), _p(_f32c(x, "x")), nseq, ctx, W, _p(out), _stream())
    _lib.check(rc, "mil_gather_eot")
    return out


# --------------------------------------------------------------------------- K1 bf16-storage variant (config 5)
def _bf16c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.bfloat16:
        raise _lib.MilHipError(f"{name}: expected a bfloat16 tensor on the GPU")
    return t if t.is_contiguous() else t.contiguous()


def cast_bf16(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    src = _f32c(src, "src")
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    rc = _lib.lib().mil_cast_bf16(_p(src), _p(out), src.numel(), _stream())
    _lib.check(rc, "mil_cast_bf16")
    return out


def gate_scores_fwd_bf16(x16, Wv16, bv, Wu16, bu, w, b, save_gates: bool = True, gates_bf16: bool = False, xbits=None,
                         xscale: float = 1.0):
    x16 = _bf16c(x16, "x")
    R, L = x16.shape
    scores = torch.empty(R, device=x16.device, dtype=torch.float32)
    """gates_bf16: save {V | U} rounded to bf16 (what gate_bwd_params_bf16 reads) instead of fp32."""
    gates = gates16 = None
    if save_gates and gates_bf16:
        gates16 = torch.empty((R, 2 * GATE_D), device=x16.device, dtype=torch.bfloat16)
    elif save_gates:
        gates = torch.empty((R, 2 * GATE_D), device=x16.device, dtype=torch.float32)
    rc = _lib.lib().mil_gate_scores_fwd_bf16(_p(x16), _p(_bf16c(Wv16, "Wv")), _p(bv), _p(_bf16c(Wu16, "Wu")), _p(bu), _p(w),
                                             _p(b), _p(scores), _p(gates), R, L, Wv16.shape[0], _p(gates16), _p(xbits),
                                             float(xscale), _stream())
    _lib.check(rc, "mil_gate_scores_fwd_bf16")
    return scores, (gates16 if gates_bf16 else gates)


def attn_pool_partial_bf16(x16, scores, layout: BagLayout, xbits=None, xscale: float = 1.0):
    x16 = _bf16c(x16, "x")
    R, L = x16.shape
    partials = torch.empty(layout.T * (L + 2), device=x16.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_partial_bf16(_p(x16), _p(scores), _p(layout.tile_map), layout.T, L, _p(partials), _p(xbits),
                                               float(xscale), _stream())
    _lib.check(rc, "mil_attn_pool_partial_bf16")
    return partials


def attn_pool_partial_h_bf16(x16, scores, layout: BagLayout, Wf, xbits=None, xscale: float = 1.0, mbits=None,
                             mscale: float = 1.0):
    """bf16 tile partials plus hrow [R, C] = x Wf^T (mil_attn_pool_partial_h_bf16)."""
    x16 = _bf16c(x16, "x")
    R, L = x16.shape
    C = Wf.shape[0]
    partials = torch.empty(layout.T * (L + 2), device=x16.device, dtype=torch.float32)
    hrow = torch.empty((R, C), device=x16.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_partial_h_bf16(_p(x16), _p(scores), _p(layout.tile_map), layout.T, L, _p(partials),
                                                 _p(_f32c(Wf, "Wf")), C, _p(hrow), _p(xbits), float(xscale), _p(mbits),
                                                 float(mscale), _stream())
    _lib.check(rc, "mil_attn_pool_partial_h_bf16")
    return partials, hrow


def attn_pool_bwd_bf16(x16, scores, lse, dM, cdot, layout: BagLayout, xbits=None, xscale: float = 1.0):
    x16 = _bf16c(x16, "x")
    R, L = x16.shape
    ds = torch.empty(R, device=x16.device, dtype=torch.float32)
    rc = _lib.lib().mil_attn_pool_bwd_bf16(_p(x16), _p(scores), _p(lse), _p(dM), _p(cdot), _p(layout.tile_map), layout.T, L,
                                           _p(ds), _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_attn_pool_bwd_bf16")
    return ds


def gate_bwd_params_x16(x16, gates, ds, w, dWv, dbv, dWu, dbu, dw, db, accumulate: bool = False, workspace=None, xbits=None,
                        xscale: float = 1.0):
    x16 = _bf16c(x16, "x")
    R, L = x16.shape
    need = _lib.lib().mil_gate_bwd_workspace_floats(R, L)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x16.device, dtype=torch.float32)
    rc = _lib.lib().mil_gate_bwd_params_x16(_p(x16), _p(gates), _p(ds), _p(w), R, L, GATE_D, _p(workspace),
                                            workspace.numel(), _p(dWv), _p(dbv), _p(dWu), _p(dbu), _p(dw), _p(db),
                                            1 if accumulate else 0, _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_gate_bwd_params_x16")
    return workspace


class _AddBagRow(torch.autograd.Function):
    """x[row] + o[bag of row]  (one-text-token fast path of the image->token attention)."""

    @staticmethod
    def forward(ctx, x, o, segs):
        x, o = _f32c(x, "x"), _f32c(o, "o")
        rows, E = x.shape
        out = torch.empty_like(x)
        rc = _lib.lib().mil_add_bag_row(_p(x), _p(o), _p(segs.q_bag), rows, E, _p(out), _stream())
        _lib.check(rc, "mil_add_bag_row")
        ctx.segs = segs
        return out

    @staticmethod
    def backward(ctx, g):
        segs = ctx.segs
        g = _f32c(g, "g")
        rows, E = g.shape
        do = torch.empty((segs.B, E), device=g.device, dtype=torch.float32)
        nch = (segs.Tq_max + 255) // 256
        ws = torch.empty(nch * segs.B * E, device=g.device, dtype=torch.float32) if nch > 1 else None
        rc = _lib.lib().mil_segment_colsum(_p(g), _p(segs.q_off), segs.B, segs.Tq_max, E, _p(do), _p(ws), _stream())
        _lib.check(rc, "mil_segment_colsum")
        return g, do, None


def add_bag_row(x, o, segs):
    """segs: AttnSegs whose QUERY side are the rows of x (q_bag / q_off)."""
    return _AddBagRow.apply(x, o, segs)


def gate_bwd_params_bf16(x16, gates, ds, w, dWv, dbv, dWu, dbu, dw, db, accumulate: bool = False, workspace=None, xbits=None,
                         xscale: float = 1.0):
    """Weight gradients on the bf16 MFMA (dPre and x rounded to bf16, fp32 accumulate); gates: bf16 [R, 384]."""
    x16 = _bf16c(x16, "x")
    gates = _bf16c(gates, "gates")
    R, L = x16.shape
    need = _lib.lib().mil_gate_bwd_workspace_floats_bf16(R, L)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x16.device, dtype=torch.float32)
    rc = _lib.lib().mil_gate_bwd_params_bf16(_p(x16), _p(gates), _p(ds), _p(w), R, L, GATE_D, _p(workspace),
                                             workspace.numel(), _p(dWv), _p(dbv), _p(dWu), _p(dbu), _p(dw), _p(db),
                                             1 if accumulate else 0, _p(xbits), float(xscale), _stream())
    _lib.check(rc, "mil_gate_bwd_params_bf16")
    return workspace


# --------------------------------------------------------------------------- CLIP-as-loss (aggregator_clip path)
class _ClipContrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, feat):
        out, feat = _f32c(out, "out"), _f32c(feat, "feat")
        b, E = out.shape
        F_ = feat.shape[1]
        loss = torch.empty(1, device=out.device, dtype=torch.float32)
        d_out = torch.empty_like(out)
        ws = torch.empty(((F_ + 3) // 4) * 4 + F_ * b * E, device=out.device, dtype=torch.float32)
        rc = _lib.lib().mil_clip_contrastive_loss(_p(out), _p(feat), b, F_, E, _p(loss), _p(d_out), _p(ws), _stream())
        _lib.check(rc, "mil_clip_contrastive_loss")
        ctx.save_for_backward(d_out)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (d_out,) = ctx.saved_tensors
        return d_out * g, None


def clip_contrastive_loss(out, feat):
    """CLIPloss_v1 (reference utils.py:261-284): out [b, E] bag embeddings vs frozen text features feat [b, F, E]."""
    return _ClipContrastive.apply(out, feat)


class _CosineEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, weight: float):
        x1, x2 = _f32c(x1, "x1"), _f32c(x2, "x2")
        B, E = x1.shape
        loss = torch.empty(1, device=x1.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d1 = torch.empty_like(x1) if need else None
        d2 = torch.empty_like(x2) if need else None
        rc = _lib.lib().mil_cosine_embedding_loss(_p(x1), _p(x2), B, E, float(weight) / B, _p(loss), _p(d1), _p(d2), _stream())
        _lib.check(rc, "mil_cosine_embedding_loss")
        ctx.save_for_backward(d1, d2)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d1, d2 = ctx.saved_tensors
        return d1 * g, d2 * g, None


def cosine_embedding_loss(x1, x2, weight: float = 1.0):
    """torch.nn.CosineEmbeddingLoss()(x1, x2, ones): mean_b (1 - cos(x1_b, x2_b)) - the 'textCosSim' term of the reference's
    training loop (train_ddp.py:102,325-329) between x_CT2CI and x_Pth2CI, [B, E] each (squeeze the token axis first)."""
    return _CosineEmbedding.apply(x1, x2, float(weight))


# --------------------------------------------------------------------------- one-text-token token->image attention
class _AbsorbQuery(torch.autograd.Function):
    """Qp[b][h] = scale * Wk_h^T qp[b][h]  (also the value projection's backward map).  T > 1: the B = bags x T rows are
    the T text tokens of each bag and the result is the grouped products' operand [bags, T H padded to a multiple of 32, E]
    with zero rows behind (written by the same launch: include/mil_hip.h mil_absorb_query_pad).  With `bias` [H C] a second
    result cb [bags, THp] = scale * bias_h . qp[b][h] (the score column constant of the other projection's bias)."""

    @staticmethod
    def forward(ctx, qp, Wk, H: int, T: int = 1, scale: float = 1.0, bias=None):
        qp, Wk = _f32c(qp, "qp"), _f32c(Wk, "Wk")
        B, I = qp.shape
        E = Wk.shape[1]
        THp = H if T == 1 else T * H + (-T * H) % 32
        Qp = torch.empty((B, H, E) if T == 1 else (B // T, THp, E), device=qp.device, dtype=torch.float32)
        cb = None
        if bias is not None:
            bias = _f32c(bias, "bias")
            cb = torch.empty((B // T, THp), device=qp.device, dtype=torch.float32)
        sh = _lib.shim() if T == 1 and scale == 1.0 and bias is None else None
        if sh is not None:
            sh.absorb_query(qp, Wk, H, Qp, _stream_int())
        else:
            rc = _lib.lib().mil_absorb_query_pad(_p(qp), _p(Wk), B, H, I // H, E, T, THp, scale, _p(bias), _p(Qp), _p(cb),
                                                 _stream())
            _lib.check(rc, "mil_absorb_query_pad")
        ctx.H, ctx.T, ctx.THp, ctx.scale = H, T, THp, scale
        ctx.with_bias = bias is not None
        if bias is None:
            ctx.save_for_backward(qp, Wk)
            return Qp
        ctx.save_for_backward(qp, Wk, bias)
        return Qp, cb

    @staticmethod
    def backward(ctx, dQp, dcb=None):
        qp, Wk = ctx.saved_tensors[:2]
        bias = ctx.saved_tensors[2] if ctx.with_bias else None
        B, I = qp.shape
        E = Wk.shape[1]
        dQp = _f32c(dQp, "dQp")
        dqp = torch.empty_like(qp) if ctx.needs_input_grad[0] else None
        dWk = None
        if ctx.needs_input_grad[1]:
            dWk = grad_slot(Wk)                       # straight into optim.FlatAdam's flat gradient buffer when there is one
            if dWk is None:
                dWk = torch.empty_like(Wk)
        sh = _lib.shim() if ctx.T == 1 and ctx.scale == 1.0 and bias is None else None
        if sh is not None:
            sh.absorb_query_bwd(qp, Wk, dQp, ctx.H, dqp, dWk, _stream_int())
            return dqp, dWk, None, None, None, None
        dbias = None
        if bias is not None:
            dcb = _f32c(dcb, "dcb")
            if ctx.needs_input_grad[5] and dWk is not None:
                dbias = grad_slot(bias)
                if dbias is None:
                    dbias = torch.empty_like(bias)
        rc = _lib.lib().mil_absorb_query_bwd_pad(_p(qp), _p(Wk), _p(dQp), B, ctx.H, I // ctx.H, E, ctx.T, ctx.THp, ctx.scale,
                                                 _p(bias), _p(dcb) if bias is not None else None, _p(dqp), _p(dWk), _p(dbias),
                                                 _stream())
        _lib.check(rc, "mil_absorb_query_bwd_pad")
        if bias is not None and ctx.needs_input_grad[5] and dbias is None:      # frozen weight, trainable bias: not a model case
            dbias = (qp.view(B, ctx.H, -1) * dcb.view(-1, ctx.THp)[:, :ctx.T * ctx.H].reshape(B, ctx.H, 1)).sum(0).reshape(-1) * ctx.scale
        return dqp, dWk, None, None, None, dbias


def _dkeys_buffer(keys, segs):
    """Gradient buffer of the keys for the absorbed pool's backward.  Capacity bucket (segments.FusionBucket): padding rows
    must hand ZERO upstream (their gradient feeds LayerNorm / bias sums of the layer below) - the bucket's padding tiles
    make the apply pass write those zeros itself; a device-length layout without them gets a cleared buffer."""
    if getattr(segs, "device_lengths", False) and not getattr(segs, "pad_tiles", False):
        return torch.zeros_like(keys)
    return torch.empty_like(keys)


class _AbsorbedPool(torch.autograd.Function):
    """pooled[b][h] = sum_n softmax_n(Qp[b][h] . (keys_n + pe_n) / sqrt(C)) keys_n.
    Also returns the keys unchanged (an alias): the caller hands THAT to the keys' other consumer, so both gradients
    of the keys arrive at this node and the backward folds them in one pass instead of autograd adding two [N, 512]
    tensors."""

    @staticmethod
    def forward(ctx, keys, pe, Qp, segs, C: int):
        keys_in = keys
        keys, pe, Qp = _f32c(keys, "keys"), _f32c(pe, "pe"), _f32c(Qp, "Qp")
        B, H, E = Qp.shape
        pooled = torch.empty_like(Qp)
        lse = torch.empty((B, H), device=keys.device, dtype=torch.float32)
        ws = torch.empty(max(1, segs.ntiles) * H * (E + 2), device=keys.device, dtype=torch.float32)
        rc = _lib.lib().mil_absorbed_pool_fwd(_p(keys), _p(pe), _p(Qp), _p(segs.k_off), _p(segs.tile_map),
                                              _p(segs.bag_tile_off), segs.ntiles, B, H, C, E, _p(pooled), _p(lse), _p(ws),
                                              _stream())
        _lib.check(rc, "mil_absorbed_pool_fwd")
        ctx.segs, ctx.C = segs, C
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(keys, pe, Qp, pooled, lse)
        return pooled, keys_in.view_as(keys_in)

    @staticmethod
    def backward(ctx, dpooled, dkeys_pass):
        keys, pe, Qp, pooled, lse = ctx.saved_tensors
        segs, C = ctx.segs, ctx.C
        B, H, E = Qp.shape
        if dpooled is None:
            return dkeys_pass, None, None, None, None
        dpooled = _f32c(dpooled, "dpooled")
        acc = _f32c(dkeys_pass, "dkeys") if dkeys_pass is not None else None
        dkeys = _dkeys_buffer(keys, segs)
        dQp = torch.empty_like(Qp)
        n_keys = keys.shape[0]
        ws = torch.empty(max(1, segs.ntiles) * H * E + 16 * n_keys, device=keys.device, dtype=torch.float32)
        rc = _lib.lib().mil_absorbed_pool_bwd(_p(keys), _p(pe), _p(Qp), _p(lse), _p(dpooled), _p(pooled), _p(segs.k_off),
                                              _p(segs.tile_map), _p(segs.bag_tile_off), segs.ntiles, n_keys, B, H, C, E,
                                              _p(acc), _p(dkeys), _p(dQp), _p(ws), _stream())
        _lib.check(rc, "mil_absorbed_pool_bwd")
        return dkeys, None, dQp, None, None


class _ValueProj(torch.autograd.Function):
    """o[b][hC + c] = Wv[hC + c] . pooled[b][h] + bv[hC + c].  T > 1: pooled is the multi-token pool's result as it stands,
    [bags, T H padded to a multiple of 32, E] (row t H + h of a bag; b = bag x T + t), and its gradient comes back in that
    layout with zero padding rows - no slice / pad copies around the node."""

    @staticmethod
    def forward(ctx, pooled, Wv, bv, H: int, T: int = 1):
        pooled, Wv = _f32c(pooled, "pooled"), _f32c(Wv, "Wv")
        E = pooled.shape[-1]
        I = Wv.shape[0]
        THp = pooled.shape[1]
        B = pooled.shape[0] * T
        if T == 1 and THp != H:
            raise ValueError("pooled must be [B, H, E]")
        o = torch.empty((B, I), device=pooled.device, dtype=torch.float32)
        rc = _lib.lib().mil_value_proj_pad(_p(pooled), _p(Wv), _p(_f32c(bv, "bv")), B, H, I // H, E, T, THp, _p(o), _stream())
        _lib.check(rc, "mil_value_proj_pad")
        ctx.save_for_backward(pooled, Wv)
        ctx.bv_param = bv                   # only to look up its flat-gradient slot in backward
        ctx.H, ctx.T = H, T
        return o

    @staticmethod
    def backward(ctx, do):
        pooled, Wv = ctx.saved_tensors
        dpooled, dWv, dbv = _value_proj_bwd(_f32c(do, "do"), Wv, ctx.bv_param, pooled, ctx.H, ctx.T)
        return dpooled, dWv, dbv, None, None


def _value_proj_bwd(do, Wv, bv, pooled, H: int = 0, T: int = 1):
    """(dpooled, dWv, dbv) of o = Wv pooled + bv in ONE launch (mil_value_proj_bwd); parameter gradients go straight into
    their flat-buffer slots when there are any."""
    E = pooled.shape[-1]
    H = H or pooled.shape[1]
    THp = pooled.shape[1]
    B = pooled.shape[0] * T
    I = Wv.shape[0]
    dpooled = torch.empty_like(pooled)
    dWv = grad_slot(Wv)
    if dWv is None:
        dWv = torch.empty_like(Wv)
    dbv = grad_slot(bv)
    if dbv is None:
        dbv = torch.empty(I, device=do.device, dtype=torch.float32)
    if E != 512 or B > SMALL_ROWS or T > 1:
        # many rows (T text tokens per bag: B = bags x T): the one-launch form walks the rows of its bias role in turn
        rc = _lib.lib().mil_absorb_query_pad(_p(do), _p(Wv), B, H, I // H, E, T, THp, 1.0, None, _p(dpooled), None, _stream())
        _lib.check(rc, "mil_absorb_query_pad")
        rc = _lib.lib().mil_absorb_query_bwd_pad(_p(do), _p(Wv), _p(pooled), B, H, I // H, E, T, THp, 1.0, None, None, None,
                                                 _p(dWv), None, _stream())
        _lib.check(rc, "mil_absorb_query_bwd_pad")
        return dpooled, dWv, colsum(do, out=dbv)
    sh = _lib.shim()
    if sh is not None:
        sh.value_proj_bwd(do, Wv, pooled, dpooled, dWv, dbv, _stream_int())
        return dpooled, dWv, dbv
    rc = _lib.lib().mil_value_proj_bwd(_p(do), _p(Wv), _p(pooled), B, H, I // H, E, _p(dpooled), _p(dWv), _p(dbv), _stream())
    _lib.check(rc, "mil_value_proj_bwd")
    return dpooled, dWv, dbv


class _AbsorbedPoolValue(torch.autograd.Function):
    """_AbsorbedPool followed by _ValueProj as ONE node: o[b] = Wv pooled[b] + bv comes out of the pool's merge launch
    (mil_absorbed_pool_value_fwd) and the backward is four launches - value projection backward (dpooled, dWv, dbv), the
    per-row dots (which form the softmax constant themselves), the apply pass, the merge - where the two nodes took eight.
    Returns (o [B, H C], keys alias)."""

    @staticmethod
    def forward(ctx, keys, pe, Qp, Wv, bv, segs, C: int):
        keys_in = keys
        keys, pe, Qp, Wv = _f32c(keys, "keys"), _f32c(pe, "pe"), _f32c(Qp, "Qp"), _f32c(Wv, "Wv")
        B, H, E = Qp.shape
        pooled = torch.empty_like(Qp)
        lse = torch.empty((B, H), device=keys.device, dtype=torch.float32)
        o = torch.empty((B, Wv.shape[0]), device=keys.device, dtype=torch.float32)
        ws = torch.empty(max(1, segs.ntiles) * H * (E + 2), device=keys.device, dtype=torch.float32)
        rc = _lib.lib().mil_absorbed_pool_value_fwd(_p(keys), _p(pe), _p(Qp), _p(segs.k_off), _p(segs.tile_map),
                                                    _p(segs.bag_tile_off), segs.ntiles, B, H, C, E, _p(Wv), _p(_f32c(bv, "bv")),
                                                    _p(pooled), _p(lse), _p(o), _p(ws), _stream())
        _lib.check(rc, "mil_absorbed_pool_value_fwd")
        ctx.segs, ctx.C, ctx.bv_param = segs, C, bv
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(keys, pe, Qp, pooled, lse, Wv)
        return o, keys_in.view_as(keys_in)

    @staticmethod
    def backward(ctx, do, dkeys_pass):
        keys, pe, Qp, pooled, lse, Wv = ctx.saved_tensors
        segs, C = ctx.segs, ctx.C
        B, H, E = Qp.shape
        if do is None:
            return dkeys_pass, None, None, None, None, None, None
        dpooled, dWv, dbv = _value_proj_bwd(_f32c(do, "do"), Wv, ctx.bv_param, pooled)
        acc = _f32c(dkeys_pass, "dkeys") if dkeys_pass is not None else None
        dkeys = _dkeys_buffer(keys, segs)
        dQp = torch.empty_like(Qp)
        n_keys = keys.shape[0]
        ws = torch.empty(max(1, segs.ntiles) * H * E + 16 * n_keys, device=keys.device, dtype=torch.float32)
        rc = _lib.lib().mil_absorbed_pool_bwd(_p(keys), _p(pe), _p(Qp), _p(lse), _p(dpooled), _p(pooled), _p(segs.k_off),
                                              _p(segs.tile_map), _p(segs.bag_tile_off), segs.ntiles, n_keys, B, H, C, E,
                                              _p(acc), _p(dkeys), _p(dQp), _p(ws), _stream())
        _lib.check(rc, "mil_absorbed_pool_bwd")
        return dkeys, None, dQp, dWv, dbv, None, None


class _LnbrAbsorbedPoolValue(torch.autograd.Function):
    """keys = LayerNorm(x + row[bag]) (the image->token attention of the block in front, one text token per bag:
    _LayerNormBagRow) followed by _AbsorbedPoolValue on those keys, as ONE node (mil_lnbr_absorbed_pool_value_fwd / _bwd): the
    pool's forward kernel makes the keys it reads, and the pool's rank-16 update of dkeys is added by the LayerNorm backward
    to the gradient it loads.  Returns (o [B, H C], keys [rows, E]); the keys' other consumer's gradient arrives at this node
    (dkeys) and is folded in the same pass."""

    @staticmethod
    def forward(ctx, x, row, gamma, beta, eps: float, pe, Qp, Wv, bv, segs, C: int, tail_rows: int):
        x, row, pe, Qp, Wv = _f32c(x, "x"), _f32c(row, "row"), _f32c(pe, "pe"), _f32c(Qp, "Qp"), _f32c(Wv, "Wv")
        B, H, E = Qp.shape
        rows = x.shape[0]
        y = torch.empty((rows + tail_rows, E), device=x.device, dtype=torch.float32)[:rows] if tail_rows else torch.empty_like(x)
        if not getattr(segs, "pad_tiles", False) and getattr(segs, "device_lengths", False):
            y.zero_()                                   # rows no tile covers must still hold finite values
        stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        pooled = torch.empty_like(Qp)
        lse = torch.empty((B, H), device=x.device, dtype=torch.float32)
        o = torch.empty((B, Wv.shape[0]), device=x.device, dtype=torch.float32)
        ws = torch.empty(max(1, segs.ntiles) * H * (E + 2), device=x.device, dtype=torch.float32)
        rc = _lib.lib().mil_lnbr_absorbed_pool_value_fwd(_p(x), _p(row), _p(_f32c(gamma, "gamma")), _p(_f32c(beta, "beta")),
                                                         float(eps), _p(pe), _p(Qp), _p(segs.k_off), _p(segs.tile_map),
                                                         _p(segs.bag_tile_off), segs.ntiles, B, H, C, E, _p(Wv),
                                                         _p(_f32c(bv, "bv")), _p(y), _p(stats), _p(pooled), _p(lse), _p(o),
                                                         _p(ws), _stream())
        _lib.check(rc, "mil_lnbr_absorbed_pool_value_fwd")
        ctx.segs, ctx.C, ctx.bv_param, ctx.beta_param = segs, C, bv, beta
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, row, gamma, stats, y, pe, Qp, pooled, lse, Wv, beta)
        return o, y

    @staticmethod
    def backward(ctx, do, dy_pass):
        x, row, gamma, stats, y, pe, Qp, pooled, lse, Wv, beta = ctx.saved_tensors
        segs, C = ctx.segs, ctx.C
        B, H, E = Qp.shape
        rows = x.shape[0]
        dWv = dbv = None
        if do is None:
            dpooled = torch.zeros_like(pooled)          # the attention output has no reader: LayerNorm backward alone
        else:
            dpooled, dWv, dbv = _value_proj_bwd(_f32c(do, "do"), Wv, ctx.bv_param, pooled)
        acc = _f32c(dy_pass, "dkeys") if dy_pass is not None else None
        dx = torch.empty_like(x)
        if not getattr(segs, "pad_tiles", False) and getattr(segs, "device_lengths", False):
            dx.zero_()
        d_row = torch.empty_like(row)
        dg = grad_slot(gamma)
        if dg is None:
            dg = torch.empty(E, device=x.device, dtype=torch.float32)
        db = grad_slot(ctx.beta_param)
        if db is None:
            db = torch.empty(E, device=x.device, dtype=torch.float32)
        dQp = torch.empty_like(Qp)
        nt = max(1, segs.ntiles)
        ws = torch.empty(nt * H * E + 16 * rows + 3 * nt * E, device=x.device, dtype=torch.float32)
        rc = _lib.lib().mil_lnbr_absorbed_pool_bwd(_p(x), _p(row), _p(gamma), _p(beta), _p(stats), _p(y), _p(pe), _p(Qp), _p(lse),
                                                   _p(dpooled), _p(pooled), _p(segs.k_off), _p(segs.tile_map),
                                                   _p(segs.bag_tile_off), segs.ntiles, rows, B, H, C, E, _p(acc), _p(dx),
                                                   _p(d_row), _p(dg), _p(db), _p(dQp), _p(ws), _stream())
        _lib.check(rc, "mil_lnbr_absorbed_pool_bwd")
        return dx, d_row, dg, db, None, None, dQp, dWv, dbv, None, None, None


def lnbr_one_token_ok(x, row, gamma, beta, Wk, Wv, bv, H: int) -> bool:
    """Shapes / trainability the fused LayerNorm(x + row) -> one-token attention node is built for."""
    return (x.dim() == 2 and x.shape[1] == 512 and x.shape[0] > 64 and H == 8 and Wk.shape[1] == 512
            and Wk.shape[0] // H in (32, 64)
            and (not torch.is_grad_enabled()
                 or (gamma.requires_grad and beta.requires_grad and Wv.requires_grad and bv.requires_grad)))


def lnbr_one_token_attention(x, row, gamma, beta, eps, pe, segs, Wk, Wv, bv, H: int, qp, tail_rows: int = 0):
    """keys = LayerNorm(x + row[bag]); token->image attention (projections absorbed) of ONE text token per bag over those
    keys.  qp [B, H C]: the projected query.  Returns (attention output [B, H C] before out_proj, keys [rows, E])."""
    C = Wk.shape[0] // H
    Qp = _AbsorbQuery.apply(qp, Wk, H)
    return _LnbrAbsorbedPoolValue.apply(x, row, gamma, beta, eps, pe, Qp, Wv, bv, segs, C, tail_rows)


def one_token_attention(q_tok, keys, pe, segs, Wq, bq, Wk, Wv, bv, H: int, qp=None):
    """Token->image attention core for ONE text token per bag, projections absorbed (csrc/absorbed_attn.hip).
    q_tok [B, E] (query + its pe), keys [R, E] WITHOUT positional encoding, pe [>= max N, E].  Returns the
    pre-out_proj attention output [B, H*C] and an alias of `keys` to be used by the keys' other consumer (see
    _AbsorbedPool).  k_proj.bias does not enter (softmax-invariant).  qp: the projected query q_proj(q_tok) when the
    caller has formed it already (ops.lin_ln_lin: the projection rides in the launch that applies the LayerNorm)."""
    if qp is None:
        qp = linear_act(q_tok, Wq, bq)
    C = Wk.shape[0] // H
    Qp = _AbsorbQuery.apply(qp, Wk, H)
    if Wk.shape[1] == 512 and H == 8 and Wv.requires_grad and bv.requires_grad:
        return _AbsorbedPoolValue.apply(keys, pe, Qp, Wv, bv, segs, C)        # pool + value projection: one node
    pooled, keys_pass = _AbsorbedPool.apply(keys, pe, Qp, segs, C)
    return _ValueProj.apply(pooled, Wv, bv, H), keys_pass


# --------------------------------------------------------------------------- multi-token absorbed attention (1 < T <= 12)
def _gg(A, a_mode, B, b_mode, grp_off, G, max_rows, M, N, K, strideB, strideC, out, bias=None, stride_bias=0, residual=None,
        pad_rows: int = 0):
    nws = _lib.lib().mil_gemm_grouped_workspace_floats(a_mode, G, max_rows, M, N) if a_mode == 1 else 0
    ws = torch.empty(nws, device=A.device, dtype=torch.float32) if nws else None
    rc = _lib.lib().mil_gemm_grouped_pad(_p(A), A.stride(0), a_mode, _p(B), B.stride(-2), b_mode, _p(out), out.stride(-2),
                                         _p(grp_off), G, max_rows, M, N, K, strideB, strideC, _p(bias), stride_bias,
                                         _p(residual), residual.stride(0) if residual is not None else 0, _p(ws), nws,
                                         int(pad_rows), _stream())
    _lib.check(rc, "mil_gemm_grouped_pad")
    return out


def _gg_nt(A, B, bias, grp_off, max_rows, zero: bool = False):
    """C[rows_g] = A[rows_g] . B[g]^T + bias[g];  A [R, K], B [G, N, K], bias [G, N] or None -> [R, N].
    zero: rows outside every group (the padding rows of a capacity bucket, segments.FusionBucket) must read 0, not
    whatever the allocation holds - the grouped kernels only write the rows of their groups."""
    G, N, K = B.shape
    out = torch.empty((A.shape[0], N), device=A.device, dtype=torch.float32)
    return _gg(A, 0, B, 0, grp_off, G, max_rows, 0, N, K, N * K, 0, out, bias, N if bias is not None else 0,
               pad_rows=A.shape[0] if zero else 0)


def _gg_nn(A, B, bias, residual, grp_off, max_rows, zero: bool = False):
    """C[rows_g] = A[rows_g] . B[g] + bias + residual;  A [R, K], B [G, K, N], bias [N] shared or None.  zero: as _gg_nt."""
    G, K, N = B.shape
    out = torch.empty((A.shape[0], N), device=A.device, dtype=torch.float32)
    return _gg(A, 0, B, 1, grp_off, G, max_rows, 0, N, K, K * N, 0, out, bias, 0, residual, pad_rows=A.shape[0] if zero else 0)


def _gg_tn(A, X, grp_off, G, max_rows):
    """C[g] = A[rows_g]^T . X[rows_g];  A [R, M], X [R, N] -> [G, M, N]."""
    M, N = A.shape[1], X.shape[1]
    out = torch.empty((G, M, N), device=A.device, dtype=torch.float32)
    return _gg(A, 1, X, 1, grp_off, G, max_rows, M, N, 0, 0, M * N, out)


def _gcs_ws(G, max_rows, ld, device):
    """Workspace of the row-parallel grouped column softmax (a few long groups), or None."""
    n = _lib.lib().mil_grp_col_softmax_workspace_floats(G, max_rows, ld)
    return torch.empty(n, device=device, dtype=torch.float32) if n else None


def _seg_colsum(Y, grp_off, G, max_rows):
    N = Y.shape[1]
    out = torch.empty((G, N), device=Y.device, dtype=torch.float32)
    nch = (max_rows + 255) // 256
    ws = torch.empty(nch * G * N, device=Y.device, dtype=torch.float32) if nch > 1 else None
    rc = _lib.lib().mil_segment_colsum(_p(Y), _p(grp_off), G, max_rows, N, _p(out), _p(ws), _stream())
    _lib.check(rc, "mil_segment_colsum")
    return out


class _GroupedNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, bias, grp_off, max_rows: int):
        A, B = _f32c(A, "A"), _f32c(B, "B")
        ctx.save_for_backward(A, B, grp_off)
        ctx.max_rows, ctx.has_bias = max_rows, bias is not None
        return _gg_nt(A, B, _f32c(bias, "bias") if bias is not None else None, grp_off, max_rows)

    @staticmethod
    def backward(ctx, dC):
        A, B, grp_off = ctx.saved_tensors
        dC = _f32c(dC, "dC")
        G = B.shape[0]
        dA = _gg_nn(dC, B, None, None, grp_off, ctx.max_rows) if ctx.needs_input_grad[0] else None
        dB = _gg_tn(dC, A, grp_off, G, ctx.max_rows) if ctx.needs_input_grad[1] else None
        db = _seg_colsum(dC, grp_off, G, ctx.max_rows) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dA, dB, db, None, None


class _GroupedNN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, bias, residual, grp_off, max_rows: int):
        A, B = _f32c(A, "A"), _f32c(B, "B")
        ctx.save_for_backward(A, B, grp_off)
        ctx.max_rows, ctx.has_bias, ctx.has_res = max_rows, bias is not None, residual is not None
        return _gg_nn(A, B, _f32c(bias, "bias") if bias is not None else None,
                      _f32c(residual, "residual") if residual is not None else None, grp_off, max_rows)

    @staticmethod
    def backward(ctx, dC):
        A, B, grp_off = ctx.saved_tensors
        dC = _f32c(dC, "dC")
        G = B.shape[0]
        dA = _gg_nt(dC, B, None, grp_off, ctx.max_rows) if ctx.needs_input_grad[0] else None       # B[g] is [K, N] = [out, in]
        dB = _gg_tn(A, dC, grp_off, G, ctx.max_rows) if ctx.needs_input_grad[1] else None
        db = colsum(dC) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dA, dB, db, (dC if ctx.has_res else None), None, None


class _GroupedTN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, X, grp_off, G: int, max_rows: int):
        A, X = _f32c(A, "A"), _f32c(X, "X")
        ctx.save_for_backward(A, X, grp_off)
        ctx.max_rows = max_rows
        return _gg_tn(A, X, grp_off, G, max_rows)

    @staticmethod
    def backward(ctx, dC):
        A, X, grp_off = ctx.saved_tensors
        dC = _f32c(dC, "dC")
        dA = _gg_nt(X, dC, None, grp_off, ctx.max_rows) if ctx.needs_input_grad[0] else None        # dC[g] is [M, N] = [out, in]
        dX = _gg_nn(A, dC, None, None, grp_off, ctx.max_rows) if ctx.needs_input_grad[1] else None
        return dA, dX, None, None, None


class _GrpColSoftmax(torch.autograd.Function):
    """Softmax over the rows of each group, per column (columns >= TH are padding: zeros)."""

    @staticmethod
    def forward(ctx, S, grp_off, G: int, TH: int, max_rows: int = 0):
        """max_rows: length of the longest group (0 = unknown): picks the kernel shape that keeps a group in registers."""
        A = S if (S.is_contiguous() and S.dtype == torch.float32) else _f32c(S, "S").clone()
        if A is S:
            ctx.mark_dirty(S)              # in place: the scores are the fresh output of the product that formed them
        rc = _lib.lib().mil_grp_col_softmax_ws(_p(A), A.stride(0), _p(grp_off), G, max_rows, TH,
                                               _p(_gcs_ws(G, max_rows, A.stride(0), A.device)), _stream())
        _lib.check(rc, "mil_grp_col_softmax_ws")
        ctx.save_for_backward(A, grp_off)
        ctx.G, ctx.TH, ctx.max_rows = G, TH, max_rows
        return A

    @staticmethod
    def backward(ctx, dA):
        A, grp_off = ctx.saved_tensors
        dA = _f32c(dA, "dA")
        dS = torch.empty_like(A)
        rc = _lib.lib().mil_grp_col_softmax_bwd_ws(_p(A), _p(dA), A.stride(0), _p(grp_off), ctx.G, ctx.max_rows, ctx.TH, _p(dS),
                                                   _p(_gcs_ws(ctx.G, ctx.max_rows, A.stride(0), A.device)), _stream())
        _lib.check(rc, "mil_grp_col_softmax_bwd_ws")
        return dS, None, None, None, None


class _RowSoftmaxT(torch.autograd.Function):
    """Softmax over the T tokens of every (row, head); column t H + h."""

    @staticmethod
    def forward(ctx, S, T: int, H: int):
        A = S if (S.is_contiguous() and S.dtype == torch.float32) else _f32c(S, "S").clone()
        if A is S:
            ctx.mark_dirty(S)
        rc = _lib.lib().mil_row_softmax_t(_p(A), A.stride(0), A.shape[0], T, H, _stream())
        _lib.check(rc, "mil_row_softmax_t")
        ctx.save_for_backward(A)
        ctx.T, ctx.H = T, H
        return A

    @staticmethod
    def backward(ctx, dA):
        (A,) = ctx.saved_tensors
        dA = _f32c(dA, "dA")
        dS = torch.empty_like(A)
        rc = _lib.lib().mil_row_softmax_t_bwd(_p(A), _p(dA), A.stride(0), A.shape[0], ctx.T, ctx.H, _p(dS), _stream())
        _lib.check(rc, "mil_row_softmax_t_bwd")
        return dS, None, None


def multi_token_ok(E: int, H: int, t_lengths) -> bool:
    T = t_lengths[0] if len(t_lengths) else 0
    return E == 512 and H == 8 and 1 < T <= 12 and all(t == T for t in t_lengths)


class _MultiTokenPoolCore(torch.autograd.Function):
    """pooled[b] = softmax_rows(kin_b Qp_b^T)^T keys_b  - the three image-side stages of the multi-token token->image
    attention as ONE autograd node.  kin must be keys + (a constant): its gradient is folded into the keys' here, and the
    node also hands the keys back as an alias for their later consumers, so every contribution to d(keys) - values,
    scores, whatever arrives through the alias - is accumulated by the `residual` operand of the skinny products
    instead of [N, 512] elementwise adds of autograd (8 x 30 us per step at 32 bags x 1024 patches)."""

    @staticmethod
    def forward(ctx, keys, kin, Qp, segs, TH: int):
        keys_in = keys
        keys, kin, Qp = _f32c(keys, "keys"), _f32c(kin.detach(), "kin"), _f32c(Qp, "Qp")
        B, off, mr = segs.B, segs.k_off, segs.Tk_max
        z = getattr(segs, "device_lengths", False)                               # capacity bucket: padding rows read 0
        A = _gg_nt(kin, Qp, None, off, mr, zero=z)                               # scores [R, THp]
        rc = _lib.lib().mil_grp_col_softmax_ws(_p(A), A.stride(0), _p(off), B, mr, TH, _p(_gcs_ws(B, mr, A.stride(0), A.device)),
                                               _stream())
        _lib.check(rc, "mil_grp_col_softmax_ws")
        pooled = _gg_tn(A, keys, off, B, mr)                                     # [B, THp, E]
        ctx.segs, ctx.TH = segs, TH
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(keys, kin, Qp, A)
        return pooled, keys_in.view_as(keys_in)

    @staticmethod
    def backward(ctx, dpooled, dkeys_pass):
        keys, kin, Qp, A = ctx.saved_tensors
        segs, TH = ctx.segs, ctx.TH
        B, off, mr = segs.B, segs.k_off, segs.Tk_max
        if dpooled is None:
            return dkeys_pass, None, None, None, None
        dpooled = _f32c(dpooled, "dpooled")
        acc = _f32c(dkeys_pass, "dkeys") if dkeys_pass is not None else None
        z = getattr(segs, "device_lengths", False)
        dA = _gg_nt(keys, dpooled, None, off, mr, zero=z)                        # dA = keys . dpooled^T
        dS = torch.zeros_like(A) if z else torch.empty_like(A)
        rc = _lib.lib().mil_grp_col_softmax_bwd_ws(_p(A), _p(dA), A.stride(0), _p(off), B, mr, TH, _p(dS),
                                                   _p(_gcs_ws(B, mr, A.stride(0), A.device)), _stream())
        _lib.check(rc, "mil_grp_col_softmax_bwd_ws")
        dkeys = None
        if ctx.needs_input_grad[0]:
            dkeys = _gg_nn(A, dpooled, None, acc, off, mr, zero=z)               # values path (+ what came through the alias)
            dkeys = _gg_nn(dS, Qp, None, dkeys, off, mr, zero=z)                 # + scores path (d kin = d keys)
        dQp = _gg_tn(dS, kin, off, B, mr) if ctx.needs_input_grad[2] else None
        return dkeys, None, dQp, None, None


class _MultiTokenRowsCore(torch.autograd.Function):
    """out = softmax_T(kin Kp_b^T + cb_b) Vp_b + bo + keys  - the image->token attention with absorbed projections and its
    residual as one node; kin = keys + (a constant), so d(keys) = dout + dS Kp comes out of one product launch."""

    @staticmethod
    def forward(ctx, keys, kin, Kp, cb, Vp, bo, segs, T: int, H: int):
        keys, kin = _f32c(keys, "keys"), _f32c(kin.detach(), "kin")
        Kp, cb, Vp, bo = _f32c(Kp, "Kp"), _f32c(cb, "cb"), _f32c(Vp, "Vp"), _f32c(bo, "bo")
        off, mr = segs.q_off, segs.Tq_max
        z = getattr(segs, "device_lengths", False)                               # capacity bucket: padding rows read 0
        A = _gg_nt(kin, Kp, cb, off, mr, zero=z)
        rc = _lib.lib().mil_row_softmax_t(_p(A), A.stride(0), A.shape[0], T, H, _stream())
        _lib.check(rc, "mil_row_softmax_t")
        out = _gg_nn(A, Vp, bo, keys, off, mr, zero=z)
        ctx.segs, ctx.T, ctx.H = segs, T, H
        ctx.save_for_backward(kin, Kp, Vp, A)
        return out

    @staticmethod
    def backward(ctx, dout):
        kin, Kp, Vp, A = ctx.saved_tensors
        segs, T, H = ctx.segs, ctx.T, ctx.H
        off, mr, B = segs.q_off, segs.Tq_max, segs.B
        dout = _f32c(dout, "dout")
        z = getattr(segs, "device_lengths", False)
        dA = _gg_nt(dout, Vp, None, off, mr, zero=z)                             # [R, THp]
        dS = torch.empty_like(A)
        rc = _lib.lib().mil_row_softmax_t_bwd(_p(A), _p(dA), A.stride(0), A.shape[0], T, H, _p(dS), _stream())
        _lib.check(rc, "mil_row_softmax_t_bwd")
        dkeys = _gg_nn(dS, Kp, None, dout, off, mr, zero=z) if ctx.needs_input_grad[0] else None      # residual + scores path
        dKp = _gg_tn(dS, kin, off, B, mr) if ctx.needs_input_grad[2] else None
        dcb = _seg_colsum(dS, off, B, mr) if ctx.needs_input_grad[3] else None
        dVp = _gg_tn(A, dout, off, B, mr) if ctx.needs_input_grad[4] else None
        dbo = colsum(dout) if ctx.needs_input_grad[5] else None
        return dkeys, None, dKp, dcb, dVp, dbo, None, None, None


def multi_token_pool_attention(q_tok, keys, kin, segs, Wq, bq, Wk, Wv, bv, H: int):
    """Token -> image attention for T text tokens per bag with the K / V projections absorbed
    (model/sam/transformer.py:291-295,113-118): the image side is three skinny grouped products around a column
    softmax instead of two [N, 512] x [512, 256] projections and an attention core.  segs: queries = tokens, keys = patches.
    kin must be keys + positional rows (a constant): its gradient is folded into the keys'.
    Returns (pre-out_proj output [B * T, H * C], keys alias - later consumers of the keys must use the alias)."""
    B, T = segs.B, segs.Tq_max
    TH = T * H
    C = Wq.shape[0] // H
    qp = linear_act(q_tok, Wq, bq)
    Qp = _AbsorbQuery.apply(qp, Wk, H, T, 1.0 / C ** 0.5)                        # k_proj.bias is softmax-invariant
    pooled, keys_pass = _MultiTokenPoolCore.apply(keys, kin, Qp, segs, TH)       # [B, THp, E]
    return _ValueProj.apply(pooled, Wv, bv, H, T), keys_pass


def multi_token_rows_attention(kin, k_tok, v_tok, segs, Wq, bq, Wk, bk, Wv, bv, Wo, bo, H: int, residual=None):
    """Image -> token attention (every patch over the T text tokens of its bag, sam/transformer.py:303-307) with the
    q and out projections absorbed into T x H key vectors Wq_h^T k_th (+ the scalar bq_h . k_th) and value vectors
    Wo_h v_th.  segs: queries = patches, keys = tokens.  Returns out_proj(attention) + residual, [R, E]."""
    B, T = segs.B, segs.Tk_max
    TH = T * H
    C = Wq.shape[0] // H
    scale = 1.0 / C ** 0.5
    kp = linear_act(k_tok, Wk, bk)                                                # [B * T, H * C]
    vp = linear_act(v_tok, Wv, bv)
    Kp, cb = _AbsorbQuery.apply(kp, Wq, H, T, scale, bq)                          # cb[b, t H + h] = scale * bq_h . k_th
    Vp = _AbsorbQuery.apply(vp, Wo.t().contiguous(), H, T, 1.0)                   # Vp[t, h] = Wo[:, hC:(h+1)C] v_th
    if residual is None:
        S = _GroupedNT.apply(kin, Kp, cb, segs.q_off, segs.Tq_max)
        return _GroupedNN.apply(_RowSoftmaxT.apply(S, T, H), Vp, bo, None, segs.q_off, segs.Tq_max)
    # with the keys as residual (the block form, sam/transformer.py:303-309) kin = keys + pe: one fused node
    return _MultiTokenRowsCore.apply(residual, kin, Kp, cb, Vp, bo, segs, T, H)


# --------------------------------------------------------------------------- split-bf16 products for frozen weights (opt-in)
def split_bf16(W, pieces: int):
    """uint16 [pieces, *W.shape]: bf16 summands of a frozen fp32 weight (include/mil_hip.h: mil_split_bf16)."""
    W = _f32c(W.detach(), "W")
    out = torch.empty((pieces,) + tuple(W.shape), device=W.device, dtype=torch.uint16)
    rc = _lib.lib().mil_split_bf16(_p(W), _p(out), W.numel(), pieces, _stream())
    _lib.check(rc, "mil_split_bf16")
    return out


def gemm_split(A, Wp, bias=None, act: int = 0, residual=None, aux=None, aux_mode: int = 0):
    """act(A . W^T + bias) + residual with W given as its bf16 pieces Wp [pieces, N, K] (mil_gemm_split)."""
    A = _f32c(A, "A")
    pieces, N, K = Wp.shape
    M = A.shape[0]
    out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    rc = _lib.lib().mil_gemm_split(_p(A), A.stride(0), _p(Wp), pieces, K, _p(out), N, M, N, K, _p(bias), act, _p(residual),
                                   residual.stride(0) if residual is not None else 0, _p(aux),
                                   aux.stride(0) if aux is not None else 0, aux_mode, _stream())
    _lib.check(rc, "mil_gemm_split")
    return out


class FrozenSplit:
    """bf16 pieces of a frozen nn.Linear weight W [N, K] and of its transpose (for the dx half of the backward)."""

    def __init__(self, W, pieces: int):
        self.pieces = pieces
        self.fwd = split_bf16(W, pieces)                             # [pieces, N, K]
        self.bwd = split_bf16(W.detach().t().contiguous(), pieces)   # [pieces, K, N]
        self.key = (W.data_ptr(), W._version)


class _LinearFrozenSplit(torch.autograd.Function):
    """act(x W^T + b) + residual for a FROZEN W given as bf16 pieces: forward and dx on the split-bf16 product."""

    @staticmethod
    def forward(ctx, x, fs: FrozenSplit, b, residual):
        ctx.fs, ctx.has_res = fs, residual is not None
        return gemm_split(x, fs.fwd, bias=b, residual=_f32c(residual, "residual") if residual is not None else None)

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy, "dy")
        dx = gemm_split(dy, ctx.fs.bwd) if ctx.needs_input_grad[0] else None
        return dx, None, None, (dy if ctx.has_res else None)


class _MlpQuickGeluFrozenSplit(torch.autograd.Function):
    """The fused MLP node (_MlpQuickGelu) on split-bf16 products; both weights frozen."""

    @staticmethod
    def forward(ctx, x, fs1: FrozenSplit, b1, fs2: FrozenSplit, b2, residual):
        x = _f32c(x, "x")
        need = ctx.needs_input_grad[0]
        pre = torch.empty((x.shape[0], fs1.fwd.shape[1]), device=x.device, dtype=torch.float32) if need else None
        h = gemm_split(x, fs1.fwd, bias=b1, act=ACT["quickgelu"], aux=pre, aux_mode=1 if need else 0)
        out = gemm_split(h, fs2.fwd, bias=b2, residual=_f32c(residual, "residual") if residual is not None else None)
        ctx.fs1, ctx.fs2, ctx.has_res = fs1, fs2, residual is not None
        ctx.save_for_backward(pre)
        return out

    @staticmethod
    def backward(ctx, dout):
        (pre,) = ctx.saved_tensors
        dout = _f32c(dout, "dout")
        dpre = gemm_split(dout, ctx.fs2.bwd, aux=pre, aux_mode=2)
        dx = gemm_split(dpre, ctx.fs1.bwd)
        return dx, None, None, None, None, (dout if ctx.has_res else None)


def linear_frozen_split(x, fs: FrozenSplit, b=None, residual=None):
    return _LinearFrozenSplit.apply(x, fs, b, residual)


def mlp_quickgelu_frozen_split(x, fs1: FrozenSplit, b1, fs2: FrozenSplit, b2, residual=None):
    return _MlpQuickGeluFrozenSplit.apply(x, fs1, b1, fs2, b2, residual)
