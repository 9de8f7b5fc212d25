"""`get_model(args)` for the image-only variant (the reference's model/utils_clip.py entry point)."""
from ._registry import build


def get_model(args):
    model = build("image_only", args)
    return model
