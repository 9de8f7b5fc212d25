"""Model factory (reference: model/utils.py:6-11)."""


def get_model(args):
    if "CT" in args.modality and "wMask" in getattr(args, "model_CT", ""):
        raise NotImplementedError("the mask-channel CT variant is outside the MIL hot path")
    from .aggregator import aggregator
    return aggregator(args)
