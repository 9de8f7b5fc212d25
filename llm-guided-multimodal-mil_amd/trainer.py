"""Fused training step for the image-only branch (BASELINE config 2/4): one bag batch ->
gate scores -> attention pool -> head -> BCE -> backward -> one flat-gradient all-reduce ->
Adam, with every buffer preallocated and every kernel enqueued through the C ABI (no autograd
graph, no host sync inside a step).

Replaces, for this branch, the reference's per-step sequence train_ddp.py:295-348
(generator(...) / criterion / zero_grad / backward / optimizer.step) and DDP's implicit bucketed
all-reduce (train_ddp.py:79): the live gradients (198 211 floats) are one contiguous buffer, so
the exchange is a single RCCL all-reduce per step."""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch
import torch.distributed as dist

from . import ops
from .bags import BagLayout
from .dist_utils import allreduce_flat, broadcast_flat

PARAM_ORDER = [
    "aggregator.attention_V.0.weight", "aggregator.attention_V.0.bias",
    "aggregator.attention_U.0.weight", "aggregator.attention_U.0.bias",
    "aggregator.attention_weights.weight", "aggregator.attention_weights.bias",
    "fc.1.weight", "fc.1.bias",
]


class FlatParams:
    """Canonical reference-named parameters as views into one flat fp32 buffer (and the same for
    gradients / Adam moments), so the optimizer and the all-reduce are single launches."""

    def __init__(self, params: Dict[str, torch.Tensor], device, order=None):
        self.order = list(order or params.keys())
        self.shapes = {k: tuple(params[k].shape) for k in self.order}
        sizes = [int(params[k].numel()) for k in self.order]
        # keep every view 16-byte aligned for float4 kernels
        self.offsets = {}
        off = 0
        for k, n in zip(self.order, sizes):
            self.offsets[k] = off
            off += (n + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        # the gradient buffer carries 4 extra floats: slot 0 of the tail holds the step's loss, so the ONE
        # all-reduce of the step also sums the loss over the ranks
        self.grad_ext = torch.zeros(off + 4, device=device, dtype=torch.float32)
        self.grad = self.grad_ext[:off]
        self.loss_slot = self.grad_ext[off:off + 1]
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        for k in self.order:
            self.view(self.flat, k).copy_(params[k])

    def view(self, buf: torch.Tensor, k: str) -> torch.Tensor:
        o = self.offsets[k]
        n = 1
        for s in self.shapes[k]:
            n *= s
        return buf[o:o + n].view(self.shapes[k])

    def p(self, k):
        return self.view(self.flat, k)

    def g(self, k):
        return self.view(self.grad, k)

    def state_dict(self):
        return {k: self.p(k).clone() for k in self.order}


class ImageOnlyTrainer:
    def __init__(self, params: Dict[str, torch.Tensor], device, lr: float = 1e-5, betas=(0.9, 0.999),
                 weight_decay: float = 1e-7, eps: float = 1e-8, world_size: int = 1, bf16_grad_mfma: bool = True):
        self.device = device
        self.fp = FlatParams(params, device, PARAM_ORDER)
        self.lr, self.betas, self.wd, self.eps = lr, betas, weight_decay, eps
        self.world = world_size
        # MIL_FORCE_COLLECTIVES=1: issue the all-reduce even at world size 1 (exercises RCCL on a one-GPU box)
        self.force_collectives = os.environ.get("MIL_FORCE_COLLECTIVES") == "1"
        # bf16 x only: weight gradient on the bf16 MFMA (dPre rounded to bf16) or on the fp32 MFMA (exact on the rounded x)
        self.bf16_grad_mfma = bf16_grad_mfma
        self.step_count = 0
        self.loss_sum = self.fp.loss_slot
        self._ws: Optional[torch.Tensor] = None
        self.last = {}
        if self.world > 1:      # DDP broadcasts rank 0's parameters at wrap time (train_ddp.py:79)
            broadcast_flat(self.fp.flat, src=0)

    # ------------------------------------------------------------------ pieces (also timed one by one by bench.py)
    def _gate_fwd(self, x, save_gates=True):
        fp = self.fp
        if x.dtype == torch.bfloat16:       # config-5 path: bf16 storage of x and of the gate weights, fp32 accumulate
            Wv16 = ops.cast_bf16(fp.p("aggregator.attention_V.0.weight"))
            Wu16 = ops.cast_bf16(fp.p("aggregator.attention_U.0.weight"))
            return ops.gate_scores_fwd_bf16(x, Wv16, fp.p("aggregator.attention_V.0.bias"), Wu16,
                                            fp.p("aggregator.attention_U.0.bias"),
                                            fp.p("aggregator.attention_weights.weight").view(-1),
                                            fp.p("aggregator.attention_weights.bias"), save_gates=save_gates)
        return ops.gate_scores_fwd(
            x, fp.p("aggregator.attention_V.0.weight"), fp.p("aggregator.attention_V.0.bias"),
            fp.p("aggregator.attention_U.0.weight"), fp.p("aggregator.attention_U.0.bias"),
            fp.p("aggregator.attention_weights.weight").view(-1), fp.p("aggregator.attention_weights.bias"),
            save_gates=save_gates)

    def forward(self, x: torch.Tensor, layout: BagLayout, y: Optional[torch.Tensor] = None,
                global_bags: Optional[int] = None):
        """Inference when y is None; with labels the fused tail also produces loss, dz, dM, cdot."""
        fp = self.fp
        scores, gates = self._gate_fwd(x, save_gates=y is not None)
        hrow = None
        if x.dtype == torch.bfloat16 and y is not None and fp.p("fc.1.weight").shape[0] <= 4:
            partials, hrow = ops.attn_pool_partial_h_bf16(x, scores, layout, fp.p("fc.1.weight"))
        elif x.dtype == torch.bfloat16:
            partials = ops.attn_pool_partial_bf16(x, scores, layout)
        elif y is not None and fp.p("fc.1.weight").shape[0] <= 4:
            # training: the pool pass also projects every patch on the head (x_i . Wf[c]); the backward then
            # needs no second pass over x (ops.attn_pool_bwd_from_h)
            partials, hrow = ops.attn_pool_partial_h(x, scores, layout, fp.p("fc.1.weight"))
        else:
            partials = ops.attn_pool_partial(x, scores, layout)
        scale = 1.0
        if y is not None:
            nb = global_bags if global_bags is not None else layout.B * self.world
            scale = 1.0 / (nb * fp.p("fc.1.weight").shape[0])
        t = ops.pool_merge_head(partials, layout, x.shape[1], fp.p("fc.1.weight"), fp.p("fc.1.bias"), y, scale,
                                scores=scores, hrow=hrow)                 # with hrow: ds comes out of this launch too
        self.last = dict(x=x, layout=layout, scores=scores, gates=gates, hrow=hrow, **t)
        return t["prob"], t["logits"]

    def backward(self):
        """Gradients of the (globally normalised) BCE loss into the flat grad buffer (overwrites it)."""
        c, fp = self.last, self.fp
        b16 = c["x"].dtype == torch.bfloat16
        # fp32 path: the head's parameter gradients ride on the gate reduce launch (below); bf16: their own launch
        head_fused = (not b16) and c["M"].shape[1] == c["x"].shape[1]
        if not head_fused:
            ops.head_bwd_params(c["dz"], c["M"], fp.g("fc.1.weight"), fp.g("fc.1.bias"), c["loss_bag"], self.loss_sum)
        if c.get("ds") is not None:
            ds = c["ds"]
        elif c.get("hrow") is not None:
            ds = ops.attn_pool_bwd_from_h(c["scores"], c["lse"], c["hrow"], c["dz"], c["cdot"], c["layout"])
        elif b16:
            ds = ops.attn_pool_bwd_bf16(c["x"], c["scores"], c["lse"], c["dM"], c["cdot"], c["layout"])
        else:
            ds, _ = ops.attn_pool_bwd(c["x"], c["scores"], c["lse"], c["dM"], c["cdot"], c["layout"], want_dx=False)
        if head_fused:
            self._ws = ops.gate_bwd_params_head(
                c["x"], c["gates"], ds, fp.p("aggregator.attention_weights.weight").view(-1),
                fp.g("aggregator.attention_V.0.weight"), fp.g("aggregator.attention_V.0.bias"),
                fp.g("aggregator.attention_U.0.weight"), fp.g("aggregator.attention_U.0.bias"),
                fp.g("aggregator.attention_weights.weight").view(-1), fp.g("aggregator.attention_weights.bias"),
                c["dz"], c["M"], fp.g("fc.1.weight"), fp.g("fc.1.bias"), c["loss_bag"], self.loss_sum, workspace=self._ws)
            return self.loss_sum
        dw_fn = ops.gate_bwd_params
        if b16:
            dw_fn = ops.gate_bwd_params_bf16 if (self.bf16_grad_mfma and c["x"].shape[1] % 256 == 0) else ops.gate_bwd_params_x16
        self._ws = dw_fn(
            c["x"], c["gates"], ds, fp.p("aggregator.attention_weights.weight").view(-1),
            fp.g("aggregator.attention_V.0.weight"), fp.g("aggregator.attention_V.0.bias"),
            fp.g("aggregator.attention_U.0.weight"), fp.g("aggregator.attention_U.0.bias"),
            fp.g("aggregator.attention_weights.weight").view(-1), fp.g("aggregator.attention_weights.bias"),
            accumulate=False, workspace=self._ws)
        return self.loss_sum

    def reduce_and_step(self):
        """One all-reduce(sum) of the flat gradient over RCCL, then Adam.  The local loss was already
        normalised by the global bag count, so the sum IS DDP's mean-of-ranks gradient."""
        if self.world > 1 or self.force_collectives:
            allreduce_flat(self.fp.grad_ext)        # gradients + loss in one collective
        self.step_count += 1
        ops.adam_step(self.fp.flat, self.fp.grad, self.fp.exp_avg, self.fp.exp_avg_sq, self.step_count, self.lr,
                      self.betas, self.eps, self.wd, 1.0)

    def train_step(self, x: torch.Tensor, layout: BagLayout, y: torch.Tensor):
        prob, z = self.forward(x, layout, y)
        self.backward()
        self.reduce_and_step()
        return self.loss_sum, prob

    # ------------------------------------------------------------------ hipGraph replay of the launch-bound part
    def capture(self, x: torch.Tensor, layout: BagLayout, y: torch.Tensor):
        """Capture forward+backward (8 launches on static buffers) into one hipGraph.  ``x`` and ``y`` become
        the static input buffers: copy new bags into them, then call ``replay_step()``.  The all-reduce and the
        Adam launch stay eager (Adam's bias corrections are per-step host scalars)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.forward(x, layout, y)
                self.backward()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self.forward(x, layout, y)
            self.backward()
        return self

    def replay_step(self):
        self._graph.replay()
        self.reduce_and_step()
        return self.loss_sum, self.last["prob"]
