"""Model factory of the image-only variant (reference: model/utils_clip.py)."""


def get_model(args):
    from .aggregator_clip import aggregator
    return aggregator(args)
