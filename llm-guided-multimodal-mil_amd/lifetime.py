"""Keep-alive bookkeeping for captured hipGraphs.

A captured graph bakes raw device pointers into its kernel nodes.  Whatever those pointers reference must stay allocated
for as long as the graph can be replayed, but several of them come out of host-side caches (BagLayout / AttnSegs tile
maps, the positional table, transposed or split copies of frozen weights) whose entries can be evicted or re-allocated.
Every cache hands its tensors to `note()`; `recording()` collects what was handed out while a graph is being warmed up
and captured, and the graph entry stores that list (graph_step.GraphedStep, trainer.ImageOnlyTrainer.capture): eviction
then only drops the cache's reference, never the memory a replay reads."""
from contextlib import contextmanager
from typing import List, Optional

_active: Optional[List[object]] = None


def note(obj):
    """Called by caches when they hand out a device-resident object."""
    if _active is not None:
        _active.append(obj)
    return obj


@contextmanager
def recording():
    global _active
    prev, _active = _active, []
    try:
        yield _active
    finally:
        if prev is not None:
            prev.extend(_active)      # nested recordings: the outer graph needs the inner objects too
        _active = prev
