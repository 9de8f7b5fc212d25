#!/usr/bin/env python3
"""Evaluation entry point (reference: test_ddp.py): load `checkpoint_best.pth.tar` strictly, eval mode, batch 1,
per-sample inference time, predictions = output[:, 1] (test_ddp.py:73,187,214-253).  The ROC / Excel reporting
of the reference is out of scope; accuracy at `--best_thres` and mean latency are printed instead."""
import os
import sys
import time

import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import mil_amd  # noqa: F401
    __package__ = "mil_amd"

from .config import create_arg_parser  # noqa: E402
from .dataset import collate_bags, load_cohort  # noqa: E402
from .train_ddp import build_model  # noqa: E402


def test(args):
    if not torch.cuda.is_available():
        raise NotImplementedError("the MIL hot path runs on MI355X only: no GPU is visible")
    dev = torch.device("cuda", int(args.gpu.split(",")[0]))
    torch.cuda.set_device(dev)
    prompts = 10 if args.CI_prompt_version == "devided" else 1
    if args.learnablePrompt:
        prompts = len(args.clinical_features) + 1
    data, args.patch_dim = load_cohort(args, "test", prompts)
    model = build_model(args).to(dev)
    if args.test_pth:
        ck = torch.load(os.path.join(args.test_pth, "checkpoint_best.pth.tar"), map_location=dev, weights_only=True)
        model.load_state_dict(ck["state_dict"])                       # strict, as test_ddp.py:99
    model.eval()
    preds, labels, times = [], [], []
    CT_SHAPE = (512, 160, 2, 2)              # synthetic stand-in for the CT encoder's feature map (aggregator.py:139-140)
    with_ct = args.variant != "image_only" and "CT" in args.modality
    inf = None
    if (getattr(args, "hip_graph", 0) and args.variant != "image_only" and not args.learnablePrompt and prompts <= 12
            and list(args.modality) in (["pathology"], ["CT", "pathology"])):
        # one forward is ~70 short launches: replay it from a hipGraph per capacity bucket (fusion_step.RaggedFusionInference)
        from .fusion_step import RaggedFusionInference
        inf = RaggedFusionInference(model, B=1, P=prompts, in_dim=args.patch_dim, ct_shape=CT_SHAPE if with_ct else None)
    cohort = None
    if inf is not None and getattr(args, "resident_cohort", 1):
        # the evaluation cohort resident in HBM too (cohort.DeviceCohort, no patch drop at test time: dataset.py:374 is
        # train-only): one feed launch per bag instead of np.load + pageable copy
        from .cohort import DeviceCohort
        lens_all = ([int(v) for v in data.lengths] if hasattr(data, "lengths") else None)
        if lens_all is None:
            import numpy as _np
            lens_all = [int(_np.load(os.path.join(data.root, k + ".npy"), mmap_mode="r").shape[0]) for k in data.keys]
        if DeviceCohort.fits(DeviceCohort.bytes_needed(lens_all, args.patch_dim), dev):
            cohort = DeviceCohort.from_dataset(data, dev, seed=args.seed, augmentation=False)
            cohort.draw_epoch(0, augment=False)
    with torch.no_grad():
        for i in range(len(data)):                                     # batch_size = 1 (test_ddp.py:73)
            if cohort is not None and not with_ct:
                n = cohort.n[i]
                slot = inf.slot(n)
                if slot.bucket.fits([n]):
                    torch.cuda.synchronize()
                    t0 = time.time()
                    ks = cohort.feed([i], slot.x, slot.bucket.len_dev, ids_dst=slot.ids)
                    out = inf.forward(slot, ks, on_device=True)
                    torch.cuda.synchronize()
                    times.append(time.time() - t0)
                    preds.append(float(out[0, 1]))
                    labels.append(int(cohort.labels[i].argmax()))
                    continue
            b = collate_bags([data[i]])
            x = b["pathology"].to(dev)
            ct = None
            if with_ct:
                from . import synthetic as syn
                ct = syn.make_ct_map(args.seed + 104729 + i, 1, CT_SHAPE[1], CT_SHAPE[2]).to(dev)
            torch.cuda.synchronize()
            t0 = time.time()
            if args.variant == "image_only":
                _, out = model([x])
            elif inf is not None:
                n = int(b["lengths"][0])
                slot = inf.slot(n)
                slot.x[:n].copy_(x[0, :n], non_blocking=True)
                if ct is not None:
                    slot.ct.copy_(ct, non_blocking=True)
                slot.ids.copy_(b["CI"], non_blocking=True)                 # the text tower is part of the replayed forward
                out = inf.forward(slot, [n])
            else:
                xs = [x] if ct is None else ([ct, x] if "pathology" in args.modality else [ct])
                out = model(xs, b["CI"].to(dev))
                out = out[0][0] if isinstance(out[0], list) else (out[0] if isinstance(out, tuple) else out)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
            preds.append(float(out[0, 1]))
            labels.append(int(b["label"][0].argmax()))
    hard = [1 if p >= args.best_thres else 0 for p in preds]
    acc = sum(int(a == b) for a, b in zip(hard, labels)) / len(labels)
    print(f"bags {len(labels)}  ACC@{args.best_thres:.4f} {acc:.4f}  Time for inference {1e3 * sum(times[1:]) / max(1, len(times) - 1):.3f} ms/bag")
    return preds, labels


if __name__ == "__main__":
    test(create_arg_parser())
