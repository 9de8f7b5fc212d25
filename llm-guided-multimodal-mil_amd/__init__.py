"""MI355X-native hot path of the LLM-guided multi-modal MIL model: gated-attention MIL
pooling, CLIP-text cross-modal fusion and the per-bag head, as hand-written HIP kernels
(gfx950) behind the reference's ``aggregator(args)`` / ``forward(x_list, x_CI)`` boundary.

Importing the package is cheap and GPU-free; the HIP library is loaded on first use by
``mil_amd._lib.lib()`` and its absence is a hard error (there is no CPU fallback)."""
from . import synthetic  # noqa: F401

__all__ = ["synthetic"]
