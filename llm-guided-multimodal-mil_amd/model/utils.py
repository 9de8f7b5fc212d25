"""`get_model(args)` for the fusion variant - the name and call signature train_ddp.py / test_ddp.py of the
reference import from model/utils.py.  The mask-channel CT model of the reference is out of scope and refused."""
from ._registry import build


def get_model(args):
    wants_mask_ct = "CT" in args.modality and "wMask" in str(getattr(args, "model_CT", ""))
    if wants_mask_ct:
        raise NotImplementedError("the mask-channel CT variant is outside the MIL hot path")
    return build("fusion", args)
