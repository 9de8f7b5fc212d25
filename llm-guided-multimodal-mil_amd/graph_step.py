"""hipGraph replay of the autograd training step (forward + loss + backward).

The authors train with ONE bag per GPU (run_train.sh:81, `--batch_size` = number of GPUs): at that size the fusion step
is a few hundred short launches and the host, not the GPU, sets the pace (measured on MI355X, 1 bag x 4096 patches:
10 frozen prompts 8.8 ms per step eager, 2.3 ms replayed; learnable prompts 10.5 -> 6.3 ms; one note 5.6 -> 1.2 ms).  `GraphedStep` captures the step body the first time a shape
signature repeats and replays it afterwards:

  * the signature is the caller's key (bag lengths, prompt count, train/eval) plus the input shapes - a step with
    another signature runs eagerly (and is captured in turn if it comes back, up to `max_graphs` graphs);
  * inputs are copied into the graph's static tensors before each replay; outputs are the graph's static tensors
    (valid until the next replay of the same graph);
  * gradients: the capture leaves every parameter's `.grad` pointing at memory the replay overwrites (FlatAdam/FlatSGD
    slots for the kernels that write in place, graph-pool tensors for the rest); they are re-attached after each
    replay, so the optimizer - which stays OUTSIDE the graph, together with the gradient all-reduce, because the
    learning-rate schedule changes its arguments - sees exactly what an eager backward would have left;
  * nothing inside the body may touch the host: the text tower has to run in its fixed-shape form
    (`CLIPText.static_rows`) or outside the body (frozen tower: pass `text_features`).
torch.cuda.graphs is the capture mechanism (hipStreamBeginCapture underneath); the kernels are the same C-ABI
launches on torch's current stream."""
from typing import Callable, Dict, List, Sequence, Tuple

import torch

from . import ops

from . import lifetime


class _Entry:
    __slots__ = ("graph", "inputs", "outputs", "grads", "keep")


class GraphedStep:
    def __init__(self, params: Sequence[torch.nn.Parameter], max_graphs: int = 4, warmup: int = 2):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.max_graphs, self.warmup = max_graphs, warmup
        self._graphs: Dict[tuple, _Entry] = {}
        self._seen: Dict[tuple, int] = {}
        self.replays = 0
        self.eager_steps = 0
        self.stream = torch.cuda.Stream()

    def _drop_grads(self):
        for p in self.params:
            p.grad = None
            p._mil_slot_used = False      # ops.grad_slot hands a parameter's flat slot out once per backward pass

    def _eager(self, inputs, body, after_backward=None):
        # Every execution of the body - eager, warm-up, capture - runs on ONE side stream: autograd binds a parameter's
        # AccumulateGrad node to the stream of the forward that created it and keeps the node while anything (a loss the
        # caller still holds, model.last_logits) references that graph; a node left over from a default-stream step
        # makes the captured backward synchronise with the default stream, which invalidates the capture.
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._drop_grads()
            out = body(*inputs)
            ops.backward(out[0])
            if after_backward is not None:
                after_backward()
            out = tuple(o.detach() for o in out)
        cur.wait_stream(self.stream)
        self.eager_steps += 1
        return out

    def _capture(self, inputs, body, after_backward=None) -> _Entry:
        ent = _Entry()
        ent.inputs = [t.clone() for t in inputs]
        cur = torch.cuda.current_stream()
        # Everything a cache hands out from here on (tile maps, segment maps, positional rows, transposed / split frozen
        # weights) is referenced by raw pointer inside the captured launches: the entry keeps those objects alive, so cache
        # eviction or a table that is re-allocated larger can never free memory a replay reads (lifetime.py).
        with lifetime.recording() as keep:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                for _ in range(self.warmup):      # caches (segment maps, positional rows, split weights) fill here
                    self._drop_grads()
                    ops.backward(body(*ent.inputs)[0])
            cur.wait_stream(self.stream)
            self._drop_grads()                    # backward inside the capture then allocates / adopts, never accumulates
            ent.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ent.graph, stream=self.stream):
                out = body(*ent.inputs)
                ops.backward(out[0])
                if after_backward is not None:      # e.g. a counted FlatAdam step: the optimizer inside the graph (the
                    after_backward()                # warm-up passes above leave it out - they must not train)
        ent.keep = keep
        ent.outputs = tuple(o.detach() for o in out)
        ent.grads = [p.grad for p in self.params]
        return ent

    def run(self, key, inputs: Sequence[torch.Tensor], body: Callable[..., Tuple[torch.Tensor, ...]], after_backward=None):
        """body(*inputs) -> (loss, *others); returns that tuple with loss.backward() done.  after_backward(): runs right
        after the backward on the same stream, and INSIDE the captured graph - for an optimizer whose step number and
        learning rate live on the device (optim.FlatAdam(counted=True)), so that a whole training step is one replay."""
        sig = (key, tuple((tuple(t.shape), t.dtype) for t in inputs))
        ent = self._graphs.get(sig)
        if ent is None:
            n = self._seen[sig] = self._seen.get(sig, 0) + 1
            if n < 2 or len(self._graphs) >= self.max_graphs:
                if len(self._seen) > 4096:
                    self._seen.clear()
                return self._eager(inputs, body, after_backward)
            ent = self._graphs[sig] = self._capture(inputs, body, after_backward)
        for dst, src in zip(ent.inputs, inputs):
            dst.copy_(src, non_blocking=True)
        ent.graph.replay()
        for p, g in zip(self.params, ent.grads):
            p.grad = g
        self.replays += 1
        return ent.outputs
