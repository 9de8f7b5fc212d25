"""Adam over ONE flat fp32 buffer for the autograd (fusion) path.

The reference builds `torch.optim.Adam(params, lr, betas=(b1, b2), weight_decay=1e-7)` (train_ddp.py:115-118) and
lets DDP all-reduce per-bucket gradients (train_ddp.py:79,347).  Here the trainable parameters are re-seated as
views of one contiguous buffer, so that per step there is
  * one gather of the gradients autograd produced into a flat gradient buffer (torch._foreach_copy_),
  * one all-reduce of that buffer when world size > 1 (RCCL; the mean is folded into Adam's grad_scale),
  * one Adam launch (mil_adam_step / mil_adam_step_counted) over everything.
Backward kernels that know a parameter's slot (ops.grad_slot: the Linear layers) write the gradient there directly,
so the gather only moves what the other ops produced.  Same arithmetic as torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected, eps outside the
sqrt); a parameter whose gradient is None is skipped altogether (no moments, no weight decay), as torch.optim.Adam
skips it: the update runs over the contiguous live ranges of the flat buffer.

Deviations from torch.optim.Adam, both deliberate: (1) ONE step number serves the whole buffer, where torch keeps a step
per parameter - a parameter that first receives a gradient at step k > 1 gets step-k bias corrections here and step-1
corrections there (the modules of this model either always or never receive gradients, so the two agree on every run the
reference can do); (2) the `_mil_zero_grad` mark (model/sam/transformer.py: a fast path left a parameter out of the graph
whose upstream gradient is a dense zero) is sticky by design: a replayed hipGraph does not re-run the Python forward that
sets it, so zero_grad() must not clear it."""
from typing import Iterable, List

import torch
import torch.distributed as dist

from . import ops


class FlatAdam:
    needs_moments = True

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-7, world_size: int = 1, counted: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdam: parameters must be fp32 on one device")
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4                     # 16-byte aligned slots (vector loads, MFMA operand rows)
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n if self.needs_moments else 0, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n if self.needs_moments else 0, device=dev, dtype=torch.float32)
        self._pviews, self._gviews = [], []
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                                  # the module now computes on the flat buffer
                self._pviews.append(view)
                self._gviews.append(self.grad[off:off + p.numel()].view(p.shape))
                p._mil_grad = self._gviews[-1]                 # ops.grad_slot: backward kernels write here directly
        self._is_zero = [True] * len(self.params)              # slot known to hold zeros (never-written / re-zeroed)
        self._live = None
        self._seg_cache = {}
        self.defaults = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}
        self.param_groups = [dict(self.defaults, params=self.params)]    # lr schedulers write param_groups[0]["lr"]
        self.world = world_size
        self.step_count = 0
        self.counted = counted
        self.step_counter = torch.zeros(1, device=dev, dtype=torch.int32) if counted else None
        # counted: the learning rate lives on the device as well, so a step captured into a hipGraph follows the schedule
        # (utils.py:232-241 writes param_groups[0]["lr"] every epoch); sync_lr() pushes a changed value before a replay
        self.lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32) if (counted and self.needs_moments) else None
        self._lr_on_dev = float(lr)
        self._done = torch.zeros(1, device=dev, dtype=torch.int32) if counted else None     # sign-off word of the one-launch update

    def zero_grad(self, set_to_none: bool = True):
        """Gradients are dropped (autograd then hands over fresh tensors without an accumulate kernel)."""
        for p in self.params:
            p.grad = None
            p._mil_slot_used = False          # ops.grad_slot: the slot may be written in place once per backward pass

    @torch.no_grad()
    def gather(self):
        """Bring every gradient into the flat buffer: nothing to do for those a backward kernel already wrote in
        place (p.grad IS the slot), one foreach copy for the rest, zeros for parameters without a gradient."""
        dst, src = [], []
        live = []
        for i, (slot, p) in enumerate(zip(self._gviews, self.params)):
            g = p.grad
            # live: received a gradient, or was left out of the graph by a fast path although upstream it carries a dense
            # zero gradient (model/sam/transformer.py: _zero_grad_params) - then the zero slot IS its gradient
            live.append(g is not None or getattr(p, "_mil_zero_grad", False))
            if g is None:
                if not self._is_zero[i]:
                    slot.zero_()
                    self._is_zero[i] = True
                continue
            self._is_zero[i] = False
            if g.data_ptr() != slot.data_ptr():
                dst.append(slot)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        self._live = tuple(live)

    def _segments(self):
        """Contiguous [begin, end) ranges of the flat buffer whose parameters received a gradient this step.
        torch.optim.Adam skips a parameter whose .grad is None - no moment update, no weight decay - so modules that are
        constructed but never used (TwoWayTransformer_CT / _Both, fc_CI2CT, prompt_embedding ...: model/aggregator.py:36-70)
        keep their initial values upstream; updating them with a zero gradient would decay them through the
        Adam-normalised weight-decay term.  The update therefore runs on the live ranges only (cached per liveness
        pattern; usually one or two ranges)."""
        live = getattr(self, "_live", None)
        if live is None or all(live):
            return [(0, self.flat.numel())]
        hit = self._seg_cache.get(live)
        if hit is None:
            hit, start = [], None
            ends = self.offsets[1:] + [self.flat.numel()]
            for i, ok in enumerate(live):
                if ok and start is None:
                    start = self.offsets[i]
                if not ok and start is not None:
                    hit.append((start, self.offsets[i]))
                    start = None
            if start is not None:
                hit.append((start, ends[-1]))
            self._seg_cache[live] = hit
        return hit

    @torch.no_grad()
    def reduce(self) -> float:
        """Gather + the step's single gradient exchange; returns the scale that turns the sum into DDP's mean."""
        self.gather()
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)   # DDP averages: sum here, 1/world inside the update
            return 1.0 / self.world
        return 1.0

    def sync_lr(self):
        """Counted mode: push param_groups[0]["lr"] to its device word when it changed.  Call before replaying a graph that
        contains step(); step() calls it itself when it runs eagerly (never while a stream capture is in progress: the
        fill would be frozen into the graph with today's value)."""
        if self.lr_dev is None:
            return
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_dev and not torch.cuda.is_current_stream_capturing():
            self.lr_dev.fill_(lr)
            self._lr_on_dev = lr

    @torch.no_grad()
    def step(self):
        scale = self.reduce()
        g = self.param_groups[0]
        segs = self._segments()
        if self.counted:
            self.sync_lr()
            if len(segs) <= 8:
                # every live range and the step-counter advance in ONE launch (three Adam launches + an increment before)
                ops.adam_step_dev_segs(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, segs, self.step_counter, self.lr_dev,
                                       self._done, g["betas"], g["eps"], g["weight_decay"], scale)
            else:
                for i, (a, b) in enumerate(segs):
                    ops.adam_step_dev(self.flat[a:b], self.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.step_counter,
                                      self.lr_dev, g["betas"], g["eps"], g["weight_decay"], scale, inc=(i == len(segs) - 1))
        else:
            self.step_count += 1
            for a, b in segs:
                ops.adam_step(self.flat[a:b], self.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], self.step_count,
                              g["lr"], g["betas"], g["eps"], g["weight_decay"], scale)

    def state_dict(self):
        step = int(self.step_counter.item()) if self.counted else self.step_count
        return {"step": step, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if self.counted:
            self.step_counter.fill_(self.step_count)
        self.param_groups[0].update(sd["param_groups"][0])


class FlatSGD(FlatAdam):
    """torch.optim.SGD(params, lr, weight_decay=1e-7) - the optimizer of the learnable-prompt runs (train_ddp.py:103-108) -
    over the same flat parameter / gradient buffers: one gather, one all-reduce, one mil_sgd_step launch."""
    needs_moments = False

    def __init__(self, params, lr: float = 1e-3, weight_decay: float = 1e-7, world_size: int = 1):
        super().__init__(params, lr=lr, weight_decay=weight_decay, world_size=world_size)
        self.defaults = {"lr": lr, "weight_decay": weight_decay}
        self.param_groups = [dict(self.defaults, params=self.params)]

    @torch.no_grad()
    def step(self):
        scale = self.reduce()
        g = self.param_groups[0]
        for a, b in self._segments():            # parameters without a gradient are skipped, as torch.optim.SGD does
            ops.sgd_step(self.flat[a:b], self.grad[a:b], g["lr"], g["weight_decay"], scale)

    def state_dict(self):
        return {"param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.param_groups[0].update(sd["param_groups"][0])
