"""A side stream for backward work that is OFF the critical path.

The token side of the two-way transformer (model/sam/transformer.py:278-309 upstream) is a chain of ~20 Linear layers on a
[B, 512] activation; in the backward pass every layer needs its input gradient before the next one can start, but nobody
waits for its weight gradient until the optimizer runs.  The few-rows backward kernel forms both in one launch
(csrc/small_linear.hip); launched as two - dx on the current stream, dW / db on this side stream - the weight halves run
beside the chain on otherwise idle CUs instead of inside it.  The side stream is ordered after everything issued so far on
the current stream when work is handed to it, and the current stream waits for it once, at the end of the backward pass
(autograd engine callback).  Only used when the gradients go straight into optim.FlatAdam's flat buffer (ops.grad_slot), so
that no autograd accumulation kernel can touch them before the join.

MEASURED (config 3, the step replayed from a hipGraph): 1.96 ms without, 2.37 ms WITH the side stream - every fork / join
pair becomes cross-queue signalling inside the graph and costs more than the 5 us it takes out of the chain.  So it is OFF
by default (MIL_SIDE_STREAM=1 turns it on, for eager experiments); the plan that replaced it: keep the chain on one stream
and form all small weight gradients in ONE grouped launch at the end of the backward pass."""
import os
from typing import Callable, Dict, List

import torch

_streams: Dict[int, torch.cuda.Stream] = {}
_pending: List[object] = []
_armed = False


def enabled() -> bool:
    return os.environ.get("MIL_SIDE_STREAM", "0") == "1"


def _side(device: torch.device) -> torch.cuda.Stream:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _streams.get(idx)
    if s is None:
        s = _streams[idx] = torch.cuda.Stream(device=idx)
    return s


def join() -> None:
    """The current stream waits for the side work handed out since the last join."""
    global _armed
    _armed = False
    if not _pending:
        return
    cur = torch.cuda.current_stream()
    cur.wait_stream(_side(cur.device))
    _pending.clear()


def run_in_backward(fn: Callable[[], None], keep) -> None:
    """Call fn() on the side stream (ordered after the work issued so far on the current stream).  `keep`: the tensors fn's
    kernels read or write - held until the join so that the allocator cannot hand their memory out again.  Must be called
    from inside a backward pass: the join is queued as an end-of-backward callback of the autograd engine."""
    global _armed
    cur = torch.cuda.current_stream()
    s = _side(cur.device)
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        fn()
    _pending.append(keep)
    if not _armed:
        _armed = True
        torch.autograd.Variable._execution_engine.queue_callback(join)
