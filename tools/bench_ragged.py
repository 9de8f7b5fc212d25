"""One ragged bag per step (the authors' regime, run_train.sh:81 + dataset.py:366-393): bags with N ~ U[2000, 15592] patches,
lengths on the device, one hipGraph per capacity bucket.  Prints ms/step for the bucketed-graph path and for the exact-length
eager path.  Default: the image-only fused step; --fusion: aggregator(args) with one clinical note per bag (the paper's model;
frozen ViT-B/32 text embedding cached per note, as training does).

--from-host / --resident (round 4): the same regime with the INPUT SIDE in the measurement - bench.ragged_regime: the cohort
resident in HBM with the per-epoch patch drop drawn on the device and one feed launch per step (what train_ddp.py --hip_graph 1
runs), beside the step alone, round 3's pageable host loop and the pinned double-buffered fallback (one JSON object)."""
import json, os, sys, time
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper

dev = torch.device("cuda")
if "--from-host" in sys.argv or "--resident" in sys.argv:
    import bench
    kind = "ct_pth" if "--ct" in sys.argv else ("fusion" if "--fusion" in sys.argv else "image")
    print(json.dumps(bench.ragged_regime(dev, kind)))
    sys.exit(0)
fusion = "--fusion" in sys.argv
steps = 300 if not fusion else 150
rng = np.random.default_rng(0)
lens = [int(v) for v in rng.integers(2000, 15593, size=steps)]
out = {}
if not fusion:
    L = 512
    p = syn.image_only_params(1, L=L)
    xs = torch.randn((16384, L), device=dev)
    y = syn.make_labels(3, 1).to(dev)
    for mode in ("bucket_graph", "exact_eager"):
        tr = ImageOnlyTrainer(p, dev, train_mode=True, counted=True)
        st = RaggedImageOnlyStepper(tr, B=1)
        def run(n):
            if mode == "bucket_graph":
                slot = st.slot(n)
                slot.x[:n].copy_(xs[:n], non_blocking=True)      # stands for the loader's H2D copy into the bucket's buffer
                slot.y.copy_(y)
                st.step(slot, [n])
            else:
                tr.train_step(xs[:n], BagLayout.make([n], dev), y)
        for n in lens[:40]:
            run(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for n in lens:
            run(n)
        torch.cuda.synchronize()
        out[mode] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        if mode == "bucket_graph":
            out["graphs"] = sum(s.graph is not None for s in st.slots.values())
    work = "1 ragged bag/step, N~U[2000,15592] x 512, image-only train-mode step"
else:
    from mil_amd.fusion_step import RaggedFusionStepper
    from mil_amd.model.utils import get_model
    from mil_amd.optim import FlatAdam, FlatSGD
    coop = "--coop" in sys.argv                       # upstream's default --learnablePrompt 1: 10 prompts, tower in the step
    P = 10 if (coop or "--prompts10" in sys.argv) else 1
    tower_in = "--tower-in-graph" in sys.argv        # frozen tower recomputed every step INSIDE the replay (no text cache)
    with_ct = "--ct" in sys.argv                     # the authors' own run (run_train.sh:81): CT + pathology, CT-Pth-Last
    CT_SHAPE = (512, 160, 2, 2)
    if coop:
        steps = 60
        lens = lens[:steps]
    args = SimpleNamespace(modality=["CT", "pathology"] if with_ct else ["pathology"], model_pathology="ABMIL", model_CI="CLIP",
                           aggregator="ABMIL", num_classes=2, learnablePrompt=int(coop), n_ctx=8, clinical_features=["f"] * 9,
                           alignment_base="CI", model_CT="resnetMC3_18", clip_layers=12, cache_text=0)
    ct = syn.make_ct_map(5, 1, CT_SHAPE[1], CT_SHAPE[2]).to(dev) if with_ct else None
    xs = torch.randn((16384, 768), device=dev)
    ids = syn.make_token_ids(2, 1, P).to(dev)
    y = syn.make_labels(3, 1).to(dev)
    train = "--eval" not in sys.argv
    for mode in ("bucket_graph", "exact_eager"):
        torch.manual_seed(1234)
        model = get_model(args).to(dev)
        model.train(train)
        trainable = [q for q in model.parameters() if q.requires_grad]
        opt = FlatSGD(trainable, lr=1e-3, weight_decay=1e-7) if coop else FlatAdam(trainable, lr=1e-5, weight_decay=1e-7, counted=True)
        st = RaggedFusionStepper(model, opt, B=1, P=P, learnable=coop, opt_in_graph=not coop,
                                 ct_shape=CT_SHAPE if with_ct else None, loss_mult=3.0 if with_ct else 1.0, cossim=with_ct,
                                 tower_in_graph=tower_in)
        tfeat = None
        if not coop:
            with torch.no_grad():
                tfeat = model.clinic_extractor(ids)               # frozen tower: cached per note (dim1/CLIP.py cache_text)
        def run(n):
            if mode == "bucket_graph":
                slot = st.slot(n)
                slot.x[:n].copy_(xs[:n], non_blocking=True)      # the loader's H2D copy into the bucket's buffer
                slot.y.copy_(y)
                if with_ct:
                    slot.ct.copy_(ct)
                if coop or tower_in:
                    slot.ids.copy_(ids)
                else:
                    slot.text.copy_(tfeat)
                st.step(slot, [n])
            else:
                opt.zero_grad()
                xl = [ct, xs[:n].unsqueeze(0)] if with_ct else [xs[:n].unsqueeze(0)]
                out_ = model(xl, ids if (coop or tower_in) else None, text_features=None if tower_in else tfeat, labels=y,
                             loss_scale=(3.0 / 2 if with_ct else None))
                loss_ = model.last_loss
                if with_ct:
                    from mil_amd import ops as _ops
                    loss_ = loss_ + _ops.cosine_embedding_loss(out_[1].squeeze(1), out_[2].squeeze(1))
                loss_.backward()
                opt.step()
        warm = lens[:(20 if coop else 40)] if mode == "bucket_graph" else lens[:5]
        for n in warm:
            run(n)
        torch.cuda.synchronize()
        todo = lens if mode == "bucket_graph" else lens[:(20 if coop else 40)]
        t0 = time.perf_counter()
        for n in todo:
            run(n)
        torch.cuda.synchronize()
        out[mode] = round((time.perf_counter() - t0) / len(todo) * 1e3, 4)
        if mode == "bucket_graph":
            out["graphs"] = len(st.gs._graphs)
            out["replays"] = st.replays
        del model, opt, st
        torch.cuda.empty_cache()
    work = (f"1 ragged bag/step, N~U[2000,15592] x 768 + {P} prompt(s) of 77 tokens"
            + (" [text tower run every step, inside the replay]" if tower_in else "")
            + (" + CT feature map [512, 160, 2, 2] (CT + pathology, loss_point CT-Pth-Last + textCosSim: run_train.sh:81)" if with_ct else "")
            + (" (learnable context through the frozen ViT-B/32 tower, SGD)" if coop else "") + ", aggregator(args) fwd+BCE+bwd+"
            + ("SGD, " if coop else "Adam, ") + ("model.train()" if train else "model.eval()"))
out["mean_patches"] = float(np.mean(lens))
print(json.dumps({"workload": work, **out}))
