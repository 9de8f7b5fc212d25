"""Lookup table behind the two `get_model` entry points: variant name -> (module, class) of this package."""
import importlib

VARIANTS = {
    "fusion": ("aggregator", "aggregator"),            # text <-> image fusion branch (reference model/aggregator.py)
    "image_only": ("aggregator_clip", "aggregator"),   # ABMIL straight on the patches (reference model/aggregator_clip.py)
}


def build(variant: str, args):
    try:
        mod_name, cls_name = VARIANTS[variant]
    except KeyError:
        raise ValueError(f"unknown model variant {variant!r}; choose from {sorted(VARIANTS)}") from None
    module = importlib.import_module(f"{__package__}.{mod_name}")
    return getattr(module, cls_name)(args)
