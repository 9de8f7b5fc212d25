"""Weight gradients of the few-rows layers, deferred to ONE grouped launch at the end of the backward pass.

Upstream every nn.Linear of the token side (model/sam/transformer.py:278-309, sam/common.py:21-26, aggregator.py:44-68) has
its weight gradient formed inside the layer's backward node.  The chain of those nodes is the critical path of the fusion
step - each layer's input gradient is needed before the previous layer can start - while the weight gradients are only read
by the optimizer.  So the chain launches the dx half alone and queues one descriptor per layer here; an end-of-backward
callback of the autograd engine then forms every queued dW / db in a single launch (mil_linear_small_dw_grouped: grid.y =
layer).  The gradients are written straight into optim.FlatAdam's flat buffer (ops.grad_slot), which is why this is only
used when such a slot exists: no autograd accumulation kernel may read the tensor before the grouped launch has run.
MIL_DEFER_DW=0 turns it off."""
import os
from typing import List

import torch

from . import _lib

_queue: List[tuple] = []
_armed = False


def enabled() -> bool:
    return os.environ.get("MIL_DEFER_DW", "1") != "0"


def flush() -> None:
    """Form every queued weight / bias gradient (current stream); called by the engine at the end of the backward pass."""
    global _armed
    _armed = False
    flush_pending()


def flush_pending() -> None:
    """Form what is queued so far, now (the end-of-pass callback stays registered for later entries).  ops.grad_slot calls
    this when a parameter is met a second time in one pass: the slot the first use returned to autograd must hold that
    use's gradient before autograd adds the second use's to it."""
    if not _queue:
        return
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(0, len(_queue), _lib.SMALL_DW_MAX):
        chunk = _queue[i:i + _lib.SMALL_DW_MAX]
        arr = (_lib.SmallDwDesc * len(chunk))()
        for d, (dy, yv, x, dW, db, act) in zip(arr, chunk):
            M, N = dy.shape
            d.dy, d.yv, d.x = dy.data_ptr(), (yv.data_ptr() if yv is not None else None), x.data_ptr()
            d.dW, d.db = (dW.data_ptr() if dW is not None else None), (db.data_ptr() if db is not None else None)
            d.lddy, d.ldyv, d.ldx = dy.stride(0), (yv.stride(0) if yv is not None else 0), x.stride(0)
            d.lddw, d.act, d.M, d.N, d.K = x.shape[1], int(act), M, N, x.shape[1]
        rc = _lib.lib().mil_linear_small_dw_grouped(arr, len(chunk), stream)
        _lib.check(rc, "mil_linear_small_dw_grouped")
    _queue.clear()


def queue_dw(dy, yv, x, dW, db, act: int) -> None:
    """Queue dW = (dy (.) act'(yv))^T x, db = its column sums; the tensors are held until the flush.  Call from inside a
    backward pass only (the flush is an end-of-backward callback of the autograd engine)."""
    global _armed
    if not _armed:
        _queue.clear()                # leftovers of a backward pass that raised before its callback ran
        _armed = True
        torch.autograd.Variable._execution_engine.queue_callback(flush)
    _queue.append((dy, yv if act != 0 else None, x, dW, db, act))
