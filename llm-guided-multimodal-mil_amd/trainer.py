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

import ctypes

from . import _lib, ops
from .bags import BagLayout, DeviceBagLayout, bucket_rows
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
    """The image-only training step on the one-call C entry point `mil_image_only_step_run` (csrc/step.hip).

    train_mode=True is a real `model.train()` step: dropout(0.5) on the patches before the gate and the pool
    (ABMIL.py:49,59) and dropout(0.25) in front of the head (aggregator.py:129), masks drawn in-kernel from Philox keep
    bits (csrc/dropout.hip) and re-read by the backward.  train_mode=False is the eval-mode (parity) step."""

    def __init__(self, params: Dict[str, torch.Tensor], device, lr: float = 1e-5, betas=(0.9, 0.999),
                 weight_decay: float = 1e-7, eps: float = 1e-8, world_size: int = 1, bf16_grad_mfma: bool = True,
                 train_mode: bool = False, accum: int = 1, seed: int = 1234, counted: bool = False,
                 loss: Optional[str] = None):
        self.device = device
        self.fp = FlatParams(params, device, PARAM_ORDER)
        self.lr, self.betas, self.wd, self.eps = lr, betas, weight_decay, eps
        self.world = world_size
        # MIL_FORCE_COLLECTIVES=1: issue the all-reduce even at world size 1 (exercises RCCL on a one-GPU box)
        self.force_collectives = os.environ.get("MIL_FORCE_COLLECTIVES") == "1"
        # bf16 x only: weight gradient on the bf16 MFMA (dPre rounded to bf16) or on the fp32 MFMA (exact on the rounded x)
        self.bf16_grad_mfma = bf16_grad_mfma
        self.step_count = 0
        # criterion as train_ddp.py:95-98 picks it: BCELoss for <= 2 classes, CrossEntropyLoss (on the sigmoid outputs,
        # one-hot float targets) above
        C = int(params["fc.1.weight"].shape[0])
        self.loss = loss if loss is not None else ("ce" if C > 2 else "bce")
        if self.loss not in ("bce", "ce"):
            raise ValueError("loss must be 'bce' or 'ce'")
        self.loss_sum = self.fp.loss_slot
        self.train_mode = bool(train_mode)
        self.seed, self.drop_pass = int(seed), 0
        self.accum = max(1, int(accum))          # micro-batches per optimizer step (one all-reduce + Adam per `accum`)
        self._micro = 0
        # counted: the step number lives on the device (Adam's bias corrections and the dropout stream position read it),
        # so the whole step - optimizer included - is the same launch sequence every time and can be replayed from a hipGraph
        self.step_counter = torch.zeros(1, device=device, dtype=torch.int32) if counted else None
        # ... and so does the learning rate: train_ddp.py sets `tr.lr = scheduled_lr(...)` every epoch (utils.py:232-241),
        # a captured step reads it from this word instead of a by-value kernel argument frozen at capture time
        self.lr_dev = torch.full((1,), float(lr), device=device, dtype=torch.float32) if counted else None
        # sign-off word of the fold launch that applies Adam: its last workgroup advances the step counter (no increment launch)
        self.done_dev = torch.zeros(1, device=device, dtype=torch.int32) if counted else None
        self._lr_on_dev = float(lr)
        self._ws: Dict[str, torch.Tensor] = {}   # grow-only per-step state (scores, gates, partials, ...)
        self._args: Optional[_lib.ImageOnlyStep] = None
        self._args_key = None
        self._keep = None                        # tensors the current struct points at (x, y, layout)
        self._w16: Optional[Dict[str, torch.Tensor]] = None
        self._w16_version, self._param_version = -1, 0
        self._graph = None
        self.last = {}
        if self.world > 1:      # DDP broadcasts rank 0's parameters at wrap time (train_ddp.py:79)
            broadcast_flat(self.fp.flat, src=0)

    # ------------------------------------------------------------------ the step descriptor
    def _buf(self, name: str, numel: int, dtype=torch.float32) -> torch.Tensor:
        t = self._ws.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            # grow-only; a captured graph keeps its own references to the buffers it was recorded with (self._graph)
            t = self._ws[name] = torch.empty(max(1, numel), device=self.device, dtype=dtype)
        return t

    def _shadows(self):
        fp = self.fp
        if self._w16 is None:
            self._w16 = {k: torch.empty(fp.p(k).shape, device=self.device, dtype=torch.bfloat16)
                         for k in ("aggregator.attention_V.0.weight", "aggregator.attention_U.0.weight")}
        if self._w16_version != self._param_version:
            for k, t in self._w16.items():
                ops.cast_bf16(fp.p(k), out=t)
            self._w16_version = self._param_version
        return self._w16

    def _fill(self, x: torch.Tensor, layout: BagLayout, y: Optional[torch.Tensor], global_bags: Optional[int]):
        """(Re)build the C struct for this batch; cached while the same tensors / layout come back."""
        b16 = x.dtype == torch.bfloat16
        if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
            raise _lib.MilHipError("ImageOnlyTrainer: x must be a float32 / bfloat16 tensor on the MI355X (no CPU path)")
        if not x.is_contiguous():
            x = x.contiguous()
        R, L = x.shape
        if R != layout.R:
            raise _lib.MilHipError(f"ImageOnlyTrainer: x has {R} rows but the bag layout covers {layout.R}")
        fp = self.fp
        C = fp.p("fc.1.weight").shape[0]
        if fp.p("fc.1.weight").shape[1] != L:
            raise _lib.MilHipError("ImageOnlyTrainer: the head's input width must equal the patch width L")
        train = self.train_mode and y is not None
        key = (x.data_ptr(), R, L, b16, id(layout), None if y is None else y.data_ptr(), global_bags, train)
        if self._args is not None and key == self._args_key:
            return self._args
        B, T = layout.B, layout.T
        a = _lib.ImageOnlyStep()
        a.struct_bytes = ctypes.sizeof(_lib.ImageOnlyStep)
        pv = lambda t: None if t is None else t.data_ptr()      # noqa: E731
        a.x, a.y, a.tile_map, a.bag_tile_off = pv(x), pv(y), pv(layout.tile_map), pv(layout.bag_tile_off)
        a.R, a.L, a.B, a.C, a.T, a.x_bf16 = R, L, B, C, T, int(b16)
        if isinstance(layout, DeviceBagLayout):                  # lengths on the device: tile map rebuilt by the step itself
            if b16:
                raise _lib.MilHipError("device-side bag lengths are supported on the fp32 path only")
            a.bag_len_dev, a.rows_dev = pv(layout.bag_len_dev), pv(layout.rows_dev)
        nb = global_bags if global_bags is not None else B * self.world
        a.loss_kind = 1 if self.loss == "ce" else 0
        a.loss_scale = 1.0 / (max(1, nb) * (C if self.loss == "bce" else 1) * self.accum)
        names = {"Wv": "aggregator.attention_V.0.weight", "bv": "aggregator.attention_V.0.bias",
                 "Wu": "aggregator.attention_U.0.weight", "bu": "aggregator.attention_U.0.bias",
                 "w": "aggregator.attention_weights.weight", "b": "aggregator.attention_weights.bias",
                 "Wf": "fc.1.weight", "bf": "fc.1.bias"}
        for f, k in names.items():
            setattr(a, f, fp.p(k).data_ptr())
            setattr(a, "d" + f, fp.g(k).data_ptr())
        if b16:
            w16 = self._shadows()
            a.Wv16, a.Wu16 = w16[names["Wv"]].data_ptr(), w16[names["Wu"]].data_ptr()
        a.loss_out = self.loss_sum.data_ptr()
        grads = y is not None
        st = dict(scores=self._buf("scores", R), partials=self._buf("partials", T * (L + 2)),
                  M=self._buf("M", B * L), lse=self._buf("lse", B), logits=self._buf("logits", B * C),
                  prob=self._buf("prob", B * C), tail_ws=self._buf("tail_ws", _lib.lib().mil_pool_tail_workspace_floats(B)))
        if grads:
            if b16:
                need = _lib.lib().mil_gate_bwd_workspace_floats_bf16(R, L) if (self.bf16_grad_mfma and L % 256 == 0) else \
                    _lib.lib().mil_gate_bwd_workspace_floats(R, L)
            else:
                need = _lib.lib().mil_gate_bwd_workspace_floats(R, L)
            # bf16-MFMA weight gradient: gates saved as bf16 (MIL_BF16_GATES=0: the fp32-MFMA gradient on fp32 gates, for A/B)
            if b16 and os.environ.get("MIL_BF16_GATES") == "0":
                self.bf16_grad_mfma = False
            g16 = b16 and self.bf16_grad_mfma and L % 256 == 0
            st.update(gates=None if g16 else self._buf("gates", R * 2 * ops.GATE_D),
                      gates16=self._buf("gates16", R * 2 * ops.GATE_D, torch.bfloat16) if g16 else None, ds=self._buf("ds", R),
                      hrow=self._buf("hrow", R * C) if C <= 4 else None, dw_ws=self._buf("dw_ws", need),
                      Mdrop=self._buf("Mdrop", B * L), loss_bag=self._buf("loss_bag", B), dz=self._buf("dz", B * C),
                      dM=self._buf("dM", B * L), cdot=self._buf("cdot", B))
            a.dw_ws_floats = st["dw_ws"].numel()
        if train:
            st.update(xbits=self._buf("xbits", R * (L // 32), torch.int32), mbits=self._buf("mbits", B * (L // 32), torch.int32))
        for k, t in st.items():
            setattr(a, k, pv(t))
        a.train, a.bf16_grad_mfma, a.seed = int(train), int(self.bf16_grad_mfma), self.seed
        a.offset_dev = pv(self.step_counter)
        a.param_flat, a.grad_flat = fp.flat.data_ptr(), fp.grad.data_ptr()
        a.exp_avg, a.exp_avg_sq, a.n_param = fp.exp_avg.data_ptr(), fp.exp_avg_sq.data_ptr(), fp.flat.numel()
        a.adam_step_dev = pv(self.step_counter)
        a.lr_dev = pv(self.lr_dev)
        a.done_dev = pv(self.done_dev)
        a.tail_ws = st["tail_ws"].data_ptr()
        a.beta1, a.beta2, a.eps, a.weight_decay, a.grad_scale = self.betas[0], self.betas[1], self.eps, self.wd, 1.0
        self._args, self._args_key = a, key
        self._keep = (x, y, layout, dict(st), self._w16)
        # host-visible views of the step's outputs (valid until the next forward)
        self.last = dict(x=x, layout=layout, scores=st["scores"][:R], M=st["M"][:B * L].view(B, L), lse=st["lse"][:B],
                         logits=st["logits"][:B * C].view(B, C), prob=st["prob"][:B * C].view(B, C))
        if grads:
            gsaved = st["gates"] if st["gates"] is not None else st["gates16"]
            self.last.update(gates=gsaved[:R * 2 * ops.GATE_D].view(R, 2 * ops.GATE_D), ds=st["ds"][:R],
                             hrow=None if st["hrow"] is None else st["hrow"][:R * C].view(R, C),
                             dz=st["dz"][:B * C].view(B, C), dM=st["dM"][:B * L].view(B, L), cdot=st["cdot"][:B],
                             loss_bag=st["loss_bag"][:B], Mdrop=st["Mdrop"][:B * L].view(B, L))
        if train:
            self.last.update(xbits=st["xbits"][:R * (L // 32)].view(R, L // 32), mbits=st["mbits"][:B * (L // 32)].view(B, L // 32))
        return a

    def _sync_lr(self):
        """Counted mode: the scheduled learning rate goes to its device word (stream-ordered, before the next launch or
        replay that reads it)."""
        if self.lr_dev is not None and self._lr_on_dev != float(self.lr) and not torch.cuda.is_current_stream_capturing():
            self.lr_dev.fill_(float(self.lr))          # never while capturing: the fill would be frozen into the graph
            self._lr_on_dev = float(self.lr)

    def _run(self, a, stages: int, keep=None):
        self._sync_lr()
        keep = self._keep if keep is None else keep
        # plain single-segment layout with every bag a multiple of 32 rows: the pool partial pass may ride in the forward launch
        if getattr(keep[2], "aligned32", False) and (stages & _lib.STAGE_GATE_FWD) and (stages & _lib.STAGE_POOL):
            stages |= _lib.STAGE_POOL_FUSED
        a.stages = stages
        a.accumulate = int(self._micro > 0)
        a.lr = self.lr
        a.adam_step = self.step_count + 1
        # dropout stream position: (micro-batch index of this optimizer step) << 32 | pass count; in counted mode the device
        # step counter is added on the device, so a replayed graph moves on by itself
        a.offset = (self._micro << 32) | (0 if self.step_counter is not None else (self.drop_pass & 0xFFFFFFFF))
        rc = _lib.lib().mil_image_only_step_run(ctypes.byref(a), ops._stream())
        _lib.check(rc, "mil_image_only_step_run")

    # ------------------------------------------------------------------ pieces
    def forward(self, x: torch.Tensor, layout: BagLayout, y: Optional[torch.Tensor] = None,
                global_bags: Optional[int] = None):
        """Inference when y is None (eval semantics); with labels the fused tail also produces loss, dz, dM, ds and, in
        train mode, the pass draws fresh dropout masks."""
        a = self._fill(x, layout, y, global_bags)
        if a.train:
            self.drop_pass += 1
        self._run(a, _lib.STAGE_TILEMAP | _lib.STAGE_DROPBITS | _lib.STAGE_GATE_FWD | _lib.STAGE_POOL | _lib.STAGE_TAIL)
        return self.last["prob"], self.last["logits"]

    def backward(self):
        """Gradients of the (globally normalised) BCE loss into the flat grad buffer (overwrites it, or adds to it for
        the 2nd.. micro-batch of an accumulation window)."""
        if self._args is None or not self._args.y:
            raise _lib.MilHipError("ImageOnlyTrainer.backward: call forward(x, layout, y) with labels first")
        self._run(self._args, _lib.STAGE_GATE_BWD | _lib.STAGE_REDUCE)
        return self.loss_sum

    def reduce_only(self):
        """The step's single collective: one all-reduce(sum) of gradients + loss over RCCL."""
        if self.world > 1 or self.force_collectives:
            allreduce_flat(self.fp.grad_ext)

    def _adam(self, a=None, keep=None):
        a = self._args if a is None else a
        self._run(a, _lib.STAGE_ADAM, keep)
        self.step_count += 1
        self._param_version += 1
        if a.x_bf16:
            self._w16_version = self._param_version     # step.hip re-cast the shadows right after the update

    def reduce_and_step(self, a=None, keep=None):
        """One all-reduce(sum) of the flat gradient over RCCL, then Adam.  The local loss was already
        normalised by the global bag count, so the sum IS DDP's mean-of-ranks gradient."""
        self.reduce_only()
        self._adam(a, keep)

    def train_step(self, x: torch.Tensor, layout: BagLayout, y: torch.Tensor):
        """forward + backward (+ all-reduce + Adam once every `accum` calls).  World size 1, accum 1: ONE C call."""
        a = self._fill(x, layout, y, None)
        if a.train:
            self.drop_pass += 1
        last = self._micro == self.accum - 1
        fused_adam = last and not (self.world > 1 or self.force_collectives)
        self._run(a, _lib.STAGE_ALL if fused_adam else _lib.STAGE_ALL & ~_lib.STAGE_ADAM)
        if fused_adam:
            self.step_count += 1
            self._param_version += 1
            if a.x_bf16:
                self._w16_version = self._param_version
        elif last:
            self.reduce_and_step()
        self._micro = 0 if last else self._micro + 1
        return self.loss_sum, self.last["prob"]

    def reset_dropout_stream(self):
        self.drop_pass = 0

    # ------------------------------------------------------------------ checkpoints (train_ddp.py:217-244 schema)
    def model_state_dict(self, prefix_map=(("aggregator.", "extractor_pathology."),)):
        """Parameters under the MODULE's key names (model/aggregator_clip.py keeps ABMIL as `extractor_pathology`), so a
        checkpoint written by the fused step loads strictly into the model and into test_ddp.py."""
        out = {}
        for k, v in self.fp.state_dict().items():
            for a, b in prefix_map:
                if k.startswith(a):
                    k = b + k[len(a):]
                    break
            out[k] = v
        return out

    def load_model_state_dict(self, sd, prefix_map=(("extractor_pathology.", "aggregator."),)):
        for k, v in sd.items():
            for a, b in prefix_map:
                if k.startswith(a):
                    k = b + k[len(a):]
                    break
            if k in self.fp.offsets:
                self.fp.p(k).copy_(v)
        self._param_version += 1

    def optimizer_state_dict(self):
        step = int(self.step_counter.item()) if self.step_counter is not None else self.step_count
        return {"step": step, "exp_avg": self.fp.exp_avg.clone(), "exp_avg_sq": self.fp.exp_avg_sq.clone(),
                "order": list(self.fp.order), "drop_pass": self.drop_pass,
                "param_groups": [{"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd}]}

    def load_optimizer_state_dict(self, sd):
        if list(sd.get("order", self.fp.order)) != list(self.fp.order):
            raise ValueError("optimizer state was saved for another parameter order")
        self.fp.exp_avg.copy_(sd["exp_avg"])
        self.fp.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.drop_pass = int(sd.get("drop_pass", 0))
        if self.step_counter is not None:
            self.step_counter.fill_(self.step_count)

    # ------------------------------------------------------------------ measurement
    def time_pieces(self, x: torch.Tensor, layout: BagLayout, y: torch.Tensor, iters: int = 20, warm: int = 3):
        """HIP-event duration (ms) of each launch group of the step, issued from C (mil_image_only_step_time)."""
        a = self._fill(x, layout, y, None)
        self._run(a, _lib.STAGE_ALL & ~_lib.STAGE_ADAM)          # state every stage reads exists
        # train mode: the forward launch draws the keep bits itself when the shape allows (mil_gate_scores_fwd_draw), so
        # the two stages are timed together, as the step runs them
        fwd = _lib.STAGE_GATE_FWD | (_lib.STAGE_DROPBITS if a.train else 0)
        groups = [("gate_fwd", fwd), ("pool_partial", _lib.STAGE_POOL),
                  ("merge_head_loss_ds", _lib.STAGE_TAIL), ("gate_bwd_dw", _lib.STAGE_GATE_BWD),
                  ("gate_bwd_reduce_and_head_params", _lib.STAGE_REDUCE)]
        if a.x_bf16:        # the bf16 weight gradient's launch pair is one entry point: time it as a whole
            groups = [g for g in groups if g[0] not in ("gate_bwd_dw", "gate_bwd_reduce_and_head_params")]
            groups.append(("gate_bwd_dw", _lib.STAGE_GATE_BWD | _lib.STAGE_REDUCE))
        if not a.x_bf16 and getattr(layout, "aligned32", False):
            # what the step actually launches for such a batch: forward with the pool partial pass in its epilogue
            groups.append(("gate_fwd_with_pool_fused", fwd | _lib.STAGE_POOL | _lib.STAGE_POOL_FUSED))
        out = {}
        ms = ctypes.c_float(0.0)
        for name, stg in groups:
            a.accumulate = 0
            rc = _lib.lib().mil_image_only_step_time(ctypes.byref(a), stg, warm, iters, ctypes.byref(ms), ops._stream())
            _lib.check(rc, "mil_image_only_step_time")
            out[name] = float(ms.value)
        # Adam on scratch copies of the parameters and moments, so timing it does not train
        fp = self.fp
        a2 = _lib.ImageOnlyStep.from_buffer_copy(a)
        scratch = [fp.flat.clone(), fp.exp_avg.clone(), fp.exp_avg_sq.clone()]
        a2.param_flat, a2.exp_avg, a2.exp_avg_sq = (t.data_ptr() for t in scratch)
        a2.adam_step_dev, a2.adam_step, a2.lr, a2.x_bf16 = None, 1, self.lr, 0
        rc = _lib.lib().mil_image_only_step_time(ctypes.byref(a2), _lib.STAGE_ADAM, warm, iters, ctypes.byref(ms), ops._stream())
        _lib.check(rc, "mil_image_only_step_time")
        out["adam"] = float(ms.value)
        return out

    def time_step_groups(self, x: torch.Tensor, layout: BagLayout, y: torch.Tensor, iters: int = 50, warm: int = 5, rot=None):
        """HIP-event duration (ms) of each launch group INSIDE the running step (mil_image_only_step_profile): the whole
        step is executed `iters` times with an event between its groups, so each kernel is timed after its predecessor, on
        the cache state the step leaves - the figure a rocprofv3 kernel trace of the step reports.  Adam runs on scratch
        copies of the parameters and moments (same launch, the trainer's state stays put).  Returns ({group: ms}, step_ms);
        step_ms is first-to-last event, i.e. the step plus its event gaps.  rot: list of (x, y) batches of this shape the
        iterations rotate through - the cache regime of a loop over several resident batches (a single batch
        stays in the Infinity Cache from step to step, which is not what the headline loop runs in)."""
        a = self._fill(x, layout, y, None)
        self._run(a, _lib.STAGE_ALL & ~_lib.STAGE_ADAM)          # state every stage reads exists
        fp = self.fp
        a2 = _lib.ImageOnlyStep.from_buffer_copy(a)
        scratch = [fp.flat.clone(), fp.exp_avg.clone(), fp.exp_avg_sq.clone()]
        a2.param_flat, a2.exp_avg, a2.exp_avg_sq = (t.data_ptr() for t in scratch)
        a2.adam_step_dev, a2.lr_dev, a2.adam_step, a2.lr, a2.accumulate = None, None, 1, self.lr, 0
        S = _lib
        fwd = S.STAGE_TILEMAP | S.STAGE_DROPBITS | S.STAGE_GATE_FWD
        # bf16: the deep forward carries the pool pass when the batch fills the chip with 256-row workgroups (csrc/step.hip
        # falls back to two launches inside the group otherwise - the label then names what was asked for)
        fused_pool = getattr(layout, "aligned32", False) and (not a.x_bf16 or (a.L in (512, 1024) and a.R >= 256 * 256
                                                                               and bool(a.gates16) and a.C == 2
                                                                               and os.environ.get("MIL_FUSE_POOL", "1") != "0"
                                                                               and os.environ.get("MIL_FUSE_POOL16", "0") != "0"))
        groups = []
        if fused_pool:
            groups.append(("gate_fwd_with_pool_fused", fwd | S.STAGE_POOL | S.STAGE_POOL_FUSED))
        else:
            groups += [("gate_fwd", fwd), ("pool_partial", S.STAGE_POOL)]
        groups.append(("merge_head_loss_ds", S.STAGE_TAIL))
        collective = self.world > 1 or self.force_collectives
        if a.x_bf16 and collective:
            groups += [("gate_bwd_dw_reduce_head", S.STAGE_GATE_BWD | S.STAGE_REDUCE), ("adam", S.STAGE_ADAM)]
        elif a.x_bf16:      # weight gradient + its fold launch, which carries head gradients, Adam and the bf16 shadows
            groups += [("gate_bwd_dw_reduce_head_adam", S.STAGE_GATE_BWD | S.STAGE_REDUCE | S.STAGE_ADAM)]
        elif collective:
            groups += [("gate_bwd_dw", S.STAGE_GATE_BWD), ("gate_bwd_reduce_head", S.STAGE_REDUCE), ("adam", S.STAGE_ADAM)]
        else:
            groups += [("gate_bwd_dw", S.STAGE_GATE_BWD), ("gate_bwd_reduce_head_adam", S.STAGE_REDUCE | S.STAGE_ADAM)]
        masks = (ctypes.c_uint32 * len(groups))(*[m for _, m in groups])
        # several short batches, per group the MEDIAN of the batch means: one stalled iteration (a 1 ms hiccup of the box
        # inside 50 steps moved a 107 us group to 126 us once) must not reach the roofline object
        nb = 5 if iters >= 20 else 1
        per = max(1, iters // nb)
        runs = []
        nrot = len(rot) if rot else 0
        for xr, yr in (rot or []):
            if xr.shape != x.shape or xr.dtype != x.dtype or yr.shape != y.shape or not xr.is_contiguous():
                raise _lib.MilHipError("time_step_groups: every rotated batch must have the shape of the first")
        xs_arr = (ctypes.c_void_p * max(1, nrot))(*[t[0].data_ptr() for t in (rot or [])])
        ys_arr = (ctypes.c_void_p * max(1, nrot))(*[t[1].data_ptr() for t in (rot or [])])
        for b in range(nb):
            out = (ctypes.c_float * (len(groups) + 1))()
            rc = _lib.lib().mil_image_only_step_profile_rot(ctypes.byref(a2), xs_arr if nrot else None, ys_arr if nrot else None,
                                                            nrot, masks, len(groups), warm if b == 0 else 1, per, out,
                                                            ops._stream())
            _lib.check(rc, "mil_image_only_step_profile_rot")
            runs.append([float(v) for v in out])
        med = [sorted(r[i] for r in runs)[nb // 2] for i in range(len(groups) + 1)]
        return {n: med[i] for i, (n, _) in enumerate(groups)}, med[len(groups)]

    # ------------------------------------------------------------------ hipGraph replay of the launch-bound part
    def capture(self, x: torch.Tensor, layout: BagLayout, y: torch.Tensor, collective_in_graph: Optional[bool] = None):
        """Capture forward+backward (static buffers) into one hipGraph.  ``x`` and ``y`` become the static input
        buffers: copy new bags into them, then call ``replay_step()``.  With counted=True the Adam launch (device step
        counter) is inside the graph as well when no all-reduce is needed.  The graph entry keeps references to every
        tensor its launches point at (layout, workspace, bf16 shadows), so later growth of the trainer's buffers or
        eviction from BagLayout's cache cannot free memory a replay reads.

        collective_in_graph (default: MIL_GRAPH_COLLECTIVE=1, counted trainers only): at world size > 1 the captured segment
        is [forward .. fold -> all_reduce -> Adam] - the step's one RCCL collective becomes a node of the graph instead of
        an eager call between two launches, so no host round trip and no eager stream hand-over sits on the exposed
        fold -> collective -> update path.  Returns self; `self._graph["collective"]` says which form was captured."""
        if self.accum != 1:
            # the micro-batch index (dropout stream position) and the accumulate flag are launch arguments frozen at capture
            raise _lib.MilHipError("ImageOnlyTrainer.capture: gradient accumulation (accum > 1) cannot be replayed from one graph")
        a = self._fill(x, layout, y, None)
        if a.train and self.step_counter is None:
            raise _lib.MilHipError("ImageOnlyTrainer.capture in train mode needs counted=True: the dropout stream position "
                                   "must live on the device, or every replay would draw the same mask")
        collective = self.world > 1 or self.force_collectives
        if collective_in_graph is None:
            collective_in_graph = os.environ.get("MIL_GRAPH_COLLECTIVE") == "1"
        collective_in_graph = bool(collective_in_graph) and collective and self.step_counter is not None and self.accum == 1
        in_graph_adam = self.step_counter is not None and self.accum == 1 and (not collective or collective_in_graph)
        stages = _lib.STAGE_ALL if in_graph_adam else _lib.STAGE_ALL & ~_lib.STAGE_ADAM
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._run(a, _lib.STAGE_ALL & ~_lib.STAGE_ADAM)
            if collective_in_graph:
                self.reduce_only()          # RCCL sets its communicator up on first use: never inside a capture
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if collective_in_graph:
                self._run(a, _lib.STAGE_ALL & ~_lib.STAGE_ADAM)
                self.reduce_only()
                self._run(a, _lib.STAGE_ADAM)
            else:
                self._run(a, stages)
        self._graph = dict(graph=graph, keep=(self._keep, dict(self._ws)), args=a, adam=in_graph_adam, last=dict(self.last),
                           collective=collective_in_graph)
        return self

    def replay_step(self):
        g = self._graph
        self._sync_lr()
        g["graph"].replay()
        self.last = g["last"]
        if g["adam"]:
            self.step_count += 1
            self._param_version += 1
        else:
            # the optimizer stage runs on the GRAPH's own descriptor: self._args / _args_key / _keep still describe the last
            # eager batch, and _fill() must keep returning that struct for that key (a replay that overwrote
            # self._args made the next capture of another bucket record the wrong bucket's buffers)
            self.reduce_and_step(g["args"], g["keep"][0])
        return self.loss_sum, self.last["prob"]


class RaggedImageOnlyStepper:
    """The one-ragged-bag-per-GPU regime on the fused step: every batch is placed in a capacity bucket
    (bags.bucket_rows: 2 048, 3 072, 4 096, ... rows), the bag lengths go to the device, and the bucket's step - the
    same launch parameters whatever the lengths - is captured into a hipGraph the second time the bucket is seen and
    replayed from then on.  A stream of bags with 2 000 .. 15 592 patches touches seven buckets, i.e. seven graphs.

        stepper = RaggedImageOnlyStepper(trainer, B=1)
        slot = stepper.slot(n_rows)            # static input buffers of the bucket: fill slot.x[:n_rows], slot.y
        loss, prob = stepper.step(slot, [n_rows])
    """

    class Slot:
        def __init__(self, cap, B, L, C, device):
            self.cap, self.visits, self.graph = cap, 0, None
            self.x = torch.zeros((cap, L), device=device, dtype=torch.float32)      # rows beyond the batch stay zero / stale: masked
            self.y = torch.zeros((B, C), device=device, dtype=torch.float32)
            self.layout = DeviceBagLayout(cap, B, device)

    def __init__(self, trainer: ImageOnlyTrainer, B: int = 1, use_graph: bool = True):
        self.tr, self.B, self.use_graph = trainer, int(B), bool(use_graph)
        if self.use_graph and trainer.step_counter is None:
            raise ValueError("graph replay needs ImageOnlyTrainer(counted=True): step number and dropout stream on the device")
        self.slots = {}
        self.replays = self.eager_steps = 0

    def slot(self, n_rows: int) -> "RaggedImageOnlyStepper.Slot":
        cap = bucket_rows(n_rows)
        s = self.slots.get(cap)
        if s is None:
            L = self.tr.fp.p("fc.1.weight").shape[1]
            C = self.tr.fp.p("fc.1.weight").shape[0]
            s = self.slots[cap] = self.Slot(cap, self.B, L, C, self.tr.device)
        return s

    def step(self, slot: "RaggedImageOnlyStepper.Slot", lengths, on_device: bool = False):
        """on_device: the lengths already sit in slot.layout.bag_len_dev (the cohort's feed launch wrote them)."""
        tr = self.tr
        if on_device:
            slot.layout.note_lengths(lengths)
        else:
            slot.layout.set_lengths(lengths)
        slot.visits += 1
        if slot.graph is None and self.use_graph and slot.visits >= 2:
            tr.capture(slot.x, slot.layout, slot.y)                # warm-up pass + capture (two optimizer-free passes)
            slot.graph = tr._graph
        if slot.graph is not None:
            tr._graph = slot.graph
            self.replays += 1
            return tr.replay_step()
        self.eager_steps += 1
        return tr.train_step(slot.x, slot.layout, slot.y)
