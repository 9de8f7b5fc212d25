"""The authors' training regime for the FUSION model: one ragged bag per GPU, its length changing every step
(reference run_train.sh:81 `--batch_size` = number of GPUs; dataset.py:366-393 drops a random 10-20 % of a bag's
patches every epoch and bags hold 2 000 .. 15 592 patches).

`graph_step.GraphedStep` keys a captured step by the exact bag lengths, so on such a cohort nothing ever replays and the
paper's model runs eagerly - host-bound at ~190 launches per step.  `RaggedFusionStepper` does for `aggregator(args)` what
`trainer.RaggedImageOnlyStepper` does for the image-only step: every batch goes into a CAPACITY bucket (bags.bucket_rows:
2 048, 3 072, 4 096, ... patch rows), the bag lengths go to the device (segments.FusionBucket), every segment map the
step reads is rebuilt there by its first launch, and the bucket's step - forward, loss, backward AND the Adam update, whose
step number and learning rate live on the device too - is ONE hipGraph captured the second time the bucket is seen.  Padding
rows carry zero softmax weight in all four pooling sites and receive exactly zero gradient.

    stepper = RaggedFusionStepper(model, opt)          # opt = optim.FlatAdam(..., counted=True)
    slot = stepper.slot(n)                             # static inputs of the bucket: slot.x[:n], slot.text, slot.y
    loss, prob = stepper.step(slot, [n])

P text tokens per bag: 1 (`CI_prompt_version='single'`, dataset.py:479-502: the absorbed one-token kernels), or up to 12
(`'devided'`, or upstream's default learnable-prompt branch with its len(clinical_features) + 1 prompts: the multi-token
grouped products, whose outputs are cleared for the padding rows).  A frozen text tower runs outside the graph (its launch
geometry follows the notes' lengths) and hands `slot.text` its embeddings; learnable prompts run the tower inside the step
on `slot.ids` in its fixed-shape form."""
from typing import Dict, Sequence

import torch

from .bags import bucket_rows
from .graph_step import GraphedStep
from .segments import FusionBucket


class RaggedFusionStepper:
    class Slot:
        def __init__(self, cap: int, B: int, C: int, in_dim: int, device, P: int = 1, ctx_len: int = 77, ct_shape=None,
                     ct_tokens: int = 0):
            self.cap, self.B = cap, B
            # CT + pathology (aggregator.py:155-173): the CT encoder's feature map is an input of the step as well
            self.ct = torch.zeros((B,) + tuple(ct_shape), device=device, dtype=torch.float32) if ct_shape else None
            self.x = torch.zeros((cap, in_dim), device=device, dtype=torch.float32)    # rows beyond the bags: padding
            self.text = torch.zeros((B, P, 512), device=device, dtype=torch.float32)   # frozen-tower embedding per prompt
            self.ids = torch.zeros((B, P, ctx_len), device=device, dtype=torch.int64)  # learnable prompts: the token ids
            self.y = torch.zeros((B, C), device=device, dtype=torch.float32)
            self.bucket = FusionBucket(cap, B, device, P, tail=[P, ct_tokens, P] if ct_shape else None)

    def __init__(self, model, opt, B: int = 1, use_graph: bool = True, max_graphs: int = 16, in_dim: int = 768,
                 opt_in_graph: bool = True, P: int = 1, learnable: bool = False, ctx_len: int = 77, ct_shape=None,
                 loss_mult: float = 1.0, cossim: bool = False, tower_in_graph: bool = False):
        """opt_in_graph=False keeps the optimizer (and, at world size > 1, its gradient all-reduce) outside the captured
        graph, as graph_step.GraphedStep does.  P: text tokens per bag (1 = one note; 10 = `CI_prompt_version='devided'` or
        the learnable-prompt branch, whose P = len(clinical_features) + 1).  learnable=True: upstream's default
        `--learnablePrompt 1` - the text tower runs INSIDE the step on `slot.ids` in its fixed-shape form
        (`clinic_extractor.model.static_rows = True`, set here) and its context vectors are trained through it."""
        if use_graph and opt_in_graph and not getattr(opt, "counted", False):
            raise ValueError("RaggedFusionStepper: an optimizer inside the graph needs optim.FlatAdam(counted=True) "
                             "(step number and learning rate on the device)")
        self.model, self.opt, self.B, self.use_graph, self.in_dim = model, opt, int(B), bool(use_graph), int(in_dim)
        self.opt_in_graph = bool(opt_in_graph)
        self.P, self.learnable, self.ctx_len = int(P), bool(learnable), int(ctx_len)
        # ct_shape (E, D, h, w): modality ['CT', 'pathology'] - slot.ct holds the CT feature map, the multi-modal bag has
        # four segments.  loss_mult: 3 for `--loss_point CT-Pth-Last` (train_ddp.py:319-322: the criterion on three outputs,
        # one head here).  cossim: add CosineEmbeddingLoss(x_CT2CI, x_Pth2CI, 1) ('textCosSim', train_ddp.py:325-329).
        self.ct_shape = tuple(ct_shape) if ct_shape else None
        self.ct_tokens = 0
        if self.ct_shape:
            E_, D_, h_, w_ = self.ct_shape
            self.ct_tokens = D_ * h_ * w_ if getattr(model.args, "model_CT", "resnetMC3_18") == "medicalNet" else D_
        self.loss_mult, self.cossim = float(loss_mult), bool(cossim)
        # tower_in_graph (frozen prompts): the reference runs encode_text on every step's notes (model/dim1/CLIP.py:71-75).
        # Outside the graph that is ~100 eager launches per note (3-4 ms of host time against a 1 ms step); inside, in the
        # tower's fixed-shape form, it is part of the replay: fill slot.ids instead of calling encode_notes().
        self.tower_inside = self.learnable or bool(tower_in_graph)
        if self.tower_inside and use_graph:
            model.clinic_extractor.model.static_rows = True      # the tower is inside the graph: no token-dependent host work
        self.device = next(model.parameters()).device
        self.C = int(model.args.num_classes)
        self.slots: Dict[int, "RaggedFusionStepper.Slot"] = {}
        self.gs = GraphedStep(list(opt.params), max_graphs=max_graphs)
        self.eager_only = 0

    @property
    def replays(self):
        return self.gs.replays

    @property
    def eager_steps(self):
        return self.gs.eager_steps + self.eager_only

    def slot(self, total_rows: int) -> "RaggedFusionStepper.Slot":
        cap = bucket_rows(total_rows)
        s = self.slots.get(cap)
        if s is None:
            s = self.slots[cap] = self.Slot(cap, self.B, self.C, self.in_dim, self.device, self.P, self.ctx_len,
                                           self.ct_shape, self.ct_tokens)
        return s

    def encode_notes(self, slot, ids):
        """Frozen text tower on this step's notes (token ids [B, 1, ctx]) -> slot.text; outside the graph."""
        with torch.no_grad():
            slot.text.copy_(self.model.clinic_extractor(ids))
        return slot.text

    def _body(self, slot):
        m = self.model
        xs = [slot.x] if slot.ct is None else [slot.ct, slot.x]
        scale = self.loss_mult / (self.B * (1 if self.C > 2 else self.C)) if self.loss_mult != 1.0 else None
        kw = dict(labels=slot.y, bucket=slot.bucket, loss_scale=scale)
        out = m(xs, slot.ids, **kw) if self.tower_inside else m(xs, None, text_features=slot.text, **kw)
        if isinstance(out[0], list):                   # args.train_contract: ([x, x, x], [CT2CI, Pth2CI], None)
            prob, toks = out[0][0], out[1]
        else:
            prob, toks = out[0], list(out[1:])
        toks = [t_ for t_ in toks if t_ is not None]
        loss = m.last_loss
        if self.cossim and len(toks) == 2:
            from . import ops
            loss = loss + ops.cosine_embedding_loss(toks[0].squeeze(1), toks[1].squeeze(1))
        return loss, prob, m.last_logits

    def step(self, slot, lengths: Sequence[int], on_device: bool = False):
        """One training step on the bags packed in slot.x (bag b at rows [sum(lengths[:b]), +lengths[b])).  Returns
        (loss, prob, logits) - static tensors of the bucket's graph once it replays.  on_device: the lengths already sit in
        slot.bucket.len_dev (the cohort's feed launch wrote them)."""
        slot.bucket.set_lengths(lengths, on_device=on_device)
        body = lambda: self._body(slot)      # noqa: E731
        if not self.use_graph:
            self.eager_only += 1
            self.gs._drop_grads()
            out = body()
            from . import ops
            ops.backward(out[0])
            self.opt.step()
            return tuple(o.detach() for o in out)
        key = ("fusion-bucket", slot.cap, self.model.training)
        if not self.opt_in_graph:
            out = self.gs.run(key, (), body)
            self.opt.step()
            return out
        self.opt.sync_lr()                   # a changed learning rate reaches its device word before the replay reads it
        return self.gs.run(key, (), body, after_backward=self.opt.step)


class RaggedFusionInference:
    """Evaluation of the fusion model (reference test_ddp.py:187-253: eval mode, one bag per forward, per-sample inference
    time) on capacity buckets: the forward of a bucket is captured into a hipGraph the second time the bucket is seen and
    replayed afterwards - a forward is ~70 short launches, eagerly the host sets the pace.  Same slots and device-side
    segments as RaggedFusionStepper; no gradients, no optimizer.

        inf = RaggedFusionInference(model, P=1)              # model.eval()
        slot = inf.slot(n); slot.x[:n].copy_(bag); inf.encode_notes(slot, ids)
        prob = inf.forward(slot, [n])                        # [B, C], valid until the next replay of that bucket"""

    def __init__(self, model, B: int = 1, P: int = 1, in_dim: int = 768, ctx_len: int = 77, ct_shape=None, use_graph: bool = True,
                 tower_in_graph: bool = True):
        self.model, self.B, self.P, self.in_dim, self.ctx_len = model, int(B), int(P), int(in_dim), int(ctx_len)
        # the text tower inside the replayed forward (fixed-shape form, fill slot.ids) or outside (encode_notes -> slot.text)
        self.tower_inside = bool(tower_in_graph) and bool(use_graph)
        if self.tower_inside:
            model.clinic_extractor.model.static_rows = True
        self.device = next(model.parameters()).device
        self.C = int(model.args.num_classes)
        self.ct_shape = tuple(ct_shape) if ct_shape else None
        self.ct_tokens = 0
        if self.ct_shape:
            E_, D_, h_, w_ = self.ct_shape
            self.ct_tokens = D_ * h_ * w_ if getattr(model.args, "model_CT", "resnetMC3_18") == "medicalNet" else D_
        self.use_graph = bool(use_graph)
        self.slots: Dict[int, RaggedFusionStepper.Slot] = {}
        self._graphs: Dict[int, tuple] = {}
        self._seen: Dict[int, int] = {}
        self.replays = self.eager = 0
        self.stream = torch.cuda.Stream()

    def slot(self, total_rows: int):
        cap = bucket_rows(total_rows)
        s = self.slots.get(cap)
        if s is None:
            s = self.slots[cap] = RaggedFusionStepper.Slot(cap, self.B, self.C, self.in_dim, self.device, self.P, self.ctx_len,
                                                           self.ct_shape, self.ct_tokens)
        return s

    def encode_notes(self, slot, ids):
        with torch.no_grad():
            slot.text.copy_(self.model.clinic_extractor(ids))
        return slot.text

    def _body(self, slot):
        xs = [slot.x] if slot.ct is None else [slot.ct, slot.x]
        if self.tower_inside:
            out = self.model(xs, slot.ids, bucket=slot.bucket)
        else:
            out = self.model(xs, None, text_features=slot.text, bucket=slot.bucket)
        prob = out[0][0] if isinstance(out[0], list) else out[0]
        return prob

    @torch.no_grad()
    def forward(self, slot, lengths: Sequence[int], on_device: bool = False):
        slot.bucket.set_lengths(lengths, on_device=on_device)
        ent = self._graphs.get(slot.cap)
        if ent is None:
            n = self._seen[slot.cap] = self._seen.get(slot.cap, 0) + 1
            if not self.use_graph or n < 2:
                self.eager += 1
                return self._body(slot)
            from . import lifetime
            cur = torch.cuda.current_stream()
            with lifetime.recording() as keep:
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    self._body(slot)                                   # caches (positional rows, segment maps) fill here
                cur.wait_stream(self.stream)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream):
                    prob = self._body(slot)
            ent = self._graphs[slot.cap] = (g, prob, keep)
        ent[0].replay()
        self.replays += 1
        return ent[1]
