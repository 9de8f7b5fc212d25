#!/usr/bin/env python3
"""Training entry point (reference: train_ddp.py): one process per GPU, seeded model, sharded bags, BCE on the
sigmoid outputs vs one-hot labels, Adam, rank-0 checkpoints with the reference's state_dict schema.

Launch either like upstream (`--multiprocessing_distributed --gpu 0,1`: mp.spawn, tcp rendezvous,
train_ddp.py:587-624) or under torchrun (RANK/LOCAL_RANK/WORLD_SIZE in the env).  Data is synthetic
(`--synthetic [N, F, bags]`): the hospital cohort is private.

Two step implementations:
  * default: `generator = DDP(get_model(args))`, autograd through the HIP operators - the reference's loop
    (train_ddp.py:295-348) with `loss_point='Last'` semantics (:323-324);
  * `--variant image_only --fused_step`: `trainer.ImageOnlyTrainer`, no autograd graph, one flat all-reduce."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import mil_amd  # noqa: F401
    __package__ = "mil_amd"

from . import ops  # noqa: E402
from .bags import BagLayout  # noqa: E402
from .config import create_arg_parser  # noqa: E402
from .dataset import collate_bags, load_cohort  # noqa: E402
from .dist_utils import broadcast_flat, env_world, init_process_group, shard_indices  # noqa: E402
from .utils import AverageMeter, ProgressMeter, calculate_accuracy, save_checkpoint, scheduled_lr  # noqa: E402


def build_model(args):
    if args.variant == "image_only":
        from .model.utils_clip import get_model
        if not getattr(args, "patch_dim", 0):
            args.patch_dim = int(args.synthetic[1])
    else:
        from .model.utils import get_model
    return get_model(args)


def main_worker(local_rank: int, nprocs: int, args):
    if args.multiprocessing_distributed:
        world, rank = nprocs * args.world_size, args.rank * nprocs + local_rank          # train_ddp.py:58
        gpu = int(args.gpu.split(",")[local_rank])
    else:
        world, rank, gpu = env_world()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        init_process_group(args.dist_backend, args.dist_url if args.multiprocessing_distributed else "env://", world,
                           rank, dev)
    if rank != 0:
        import builtins
        builtins.print = lambda *a, **k: None                                            # train_ddp.py:45-48
    torch.manual_seed(args.seed)
    prompts = 10 if args.CI_prompt_version == "devided" else 1
    if args.learnablePrompt:
        prompts = len(args.clinical_features) + 1                                        # dim1/CLIP.py:19
    data, args.patch_dim = load_cohort(args, "train", prompts)                           # train_ddp.py:188-196
    per_gpu = max(1, args.batch_size // world)                                           # train_ddp.py:75
    model = build_model(args).to(dev)
    lr0 = 1e-3 if args.num_classes > 2 else 1e-5                                         # train_ddp.py:110-113
    args.lr = lr0

    fused = args.fused_step and args.variant == "image_only"
    if fused:
        from .trainer import ImageOnlyTrainer
        sd = model.state_dict()
        params = {k.replace("extractor_pathology.", "aggregator."): v for k, v in sd.items()}
        # a true model.train() step: dropout on the patches and in front of the head, masks drawn in-kernel (Philox)
        tr = ImageOnlyTrainer(params, dev, lr=lr0, betas=(args.b1, args.b2), weight_decay=1e-7, world_size=world,
                              train_mode=not getattr(args, "no_dropout", False), seed=args.seed,
                              counted=bool(getattr(args, "hip_graph", 0)))
        stepper = None
        if getattr(args, "hip_graph", 0):
            # ragged bags whose lengths change every step (one bag per GPU upstream): lengths on the device, one captured
            # graph per capacity bucket (trainer.RaggedImageOnlyStepper)
            from .trainer import RaggedImageOnlyStepper
            stepper = RaggedImageOnlyStepper(tr, B=per_gpu)
        if args.resume:
            ck = torch.load(args.resume, map_location=dev, weights_only=True)
            tr.load_model_state_dict(ck["state_dict"])
            tr.load_optimizer_state_dict(ck["optimizer"])
            args.start_epoch = ck["epoch"]
    else:
        generator = model
        flat_opt = bool(getattr(args, "flat_adam", 1))
        if world > 1 and not flat_opt:
            generator = torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)
        # train_ddp.py:95-98: CrossEntropyLoss above two classes (applied to the sigmoid outputs, float one-hot targets)
        criterion = torch.nn.CrossEntropyLoss() if args.num_classes > 2 else torch.nn.BCELoss()
        trainable = [p for p in model.parameters() if p.requires_grad]
        if args.learnablePrompt:                                                         # train_ddp.py:104-109
            lr0 = args.lr = 1e-3
            if flat_opt:
                from .optim import FlatSGD
                optimizer = FlatSGD(trainable, lr=lr0, weight_decay=1e-7, world_size=world)
            else:
                optimizer = torch.optim.SGD(trainable, lr=lr0, weight_decay=1e-7)
        elif flat_opt:                                                                   # train_ddp.py:110-118, flat
            from .optim import FlatAdam
            # one clinical note per bag + --hip_graph: the capacity-bucket stepper below replays the whole step, optimizer
            # included, so step number and learning rate live on the device
            bucketed = (bool(getattr(args, "hip_graph", 0)) and args.variant != "image_only" and prompts <= 12
                        and list(args.modality) in (["pathology"], ["CT", "pathology"]))
            optimizer = FlatAdam(trainable, lr=lr0, betas=(args.b1, args.b2), weight_decay=1e-7, world_size=world,
                                 counted=bucketed)
        else:
            optimizer = torch.optim.Adam(trainable, lr=lr0, betas=(args.b1, args.b2), weight_decay=1e-7)
        if flat_opt and world > 1:
            broadcast_flat(optimizer.flat, src=0)                                        # DDP's initial broadcast
        # criterion inside the module (aggregator.forward(labels=...): pool + head + loss as one fused node) whenever the
        # module is called directly - DDP's unused-parameter search needs the loss to hang off forward's outputs
        fuse_loss = generator is model and args.variant != "image_only"

        def loss_of(out_prob, y_):
            return model.last_loss if (fuse_loss and model.last_loss is not None) else criterion(out_prob, y_)

        three_terms = args.loss_point == "CT-Pth-Last" and "CT" in args.modality and "pathology" in args.modality

        def unpack(out):
            """(prob, [tokens]) from either return contract (aggregator.py:202-209 / train_ddp.py:300)."""
            if isinstance(out, tuple) and isinstance(out[0], list):
                return out[0][0], [t_ for t_ in out[1] if t_ is not None]
            if isinstance(out, tuple):
                return out[0], [t_ for t_ in out[1:] if t_ is not None]
            return out, []

        def total_loss(prob_, toks, y_):
            """train_ddp.py:318-329: 'Last' = criterion(out); 'CT-Pth-Last' sums the criterion over the three outputs (one
            head here, so three times the same term); 'textCosSim' adds the cosine term between the two text-aligned tokens."""
            fused_ = fuse_loss and model.last_loss is not None     # the factor 3 then sits in the fused node's loss_scale
            loss_ = loss_of(prob_, y_)
            if three_terms and not fused_:
                loss_ = loss_ * 3.0
            if "textCosSim" in args.loss and len(toks) == 2:
                from . import ops
                loss_ = loss_ + ops.cosine_embedding_loss(toks[0].squeeze(1), toks[1].squeeze(1))
            return loss_
        graphed = fstepper = None
        CT_SHAPE = (512, 160, 2, 2)          # synthetic stand-in for the CT encoder's feature map (aggregator.py:139-140)
        if (getattr(args, "hip_graph", 0) and flat_opt and args.variant != "image_only" and prompts <= 12
                and list(args.modality) in (["pathology"], ["CT", "pathology"])):
            # the authors' regime (one ragged bag per GPU, run_train.sh:81; a fresh patch drop every epoch,
            # dataset.py:366-393): bag lengths on the device, one graph per capacity bucket (fusion_step.py).  Adam with its
            # step number and rate on the device rides inside the graph at world size 1; the learnable-prompt runs' SGD
            # (train_ddp.py:104-109) and any multi-GPU all-reduce stay outside.
            from .fusion_step import RaggedFusionStepper
            fstepper = RaggedFusionStepper(model, optimizer, B=per_gpu, P=prompts, learnable=bool(args.learnablePrompt),
                                           opt_in_graph=(world == 1 and getattr(optimizer, "counted", False)),
                                           ct_shape=CT_SHAPE if "CT" in args.modality else None,
                                           loss_mult=3.0 if (args.loss_point == "CT-Pth-Last" and "CT" in args.modality) else 1.0,
                                           cossim="textCosSim" in args.loss,
                                           # frozen tower: inside the replay too, unless its embeddings are cached per note
                                           tower_in_graph=not args.learnablePrompt and not getattr(args, "cache_text", 0))
        if getattr(args, "hip_graph", 0):
            # replay the step body from a hipGraph once a batch shape repeats (graph_step.py); optimizer and the
            # gradient all-reduce stay outside, so this needs the flat optimizers when world > 1 (no DDP hooks)
            if world > 1 and not flat_opt:
                raise ValueError("--hip_graph with several GPUs needs --flat_adam 1")
            from .graph_step import GraphedStep
            graphed = GraphedStep(trainable)
            if args.variant != "image_only" and args.learnablePrompt:
                model.clinic_extractor.model.static_rows = True     # the tower is inside the graph: fixed-shape form
        if args.resume:
            ck = torch.load(args.resume, map_location=dev, weights_only=True)
            model.load_state_dict(ck["state_dict"])
            optimizer.load_state_dict(ck["optimizer"])
            args.start_epoch = ck["epoch"]

    # ---- the input side.  With a bucketed stepper the cohort lives in HBM (cohort.DeviceCohort): loaded once, the
    # per-epoch patch drop drawn on the device, every step fed by ONE gather launch into the bucket's static buffers -
    # instead of the reference's per-step np.load + random.sample + pad + H2D copy (dataset.py:366-393, train_ddp.py:274-293)
    cohort = None
    any_stepper = (stepper if fused else fstepper)
    if any_stepper is not None and getattr(args, "resident_cohort", 1):
        from .cohort import DeviceCohort
        lens_all = ([int(v) for v in data.lengths] if hasattr(data, "lengths") else None)
        if lens_all is None:
            import numpy as _np
            lens_all = [int(_np.load(os.path.join(data.root, k + ".npy"), mmap_mode="r").shape[0]) for k in data.keys]
        if DeviceCohort.fits(DeviceCohort.bytes_needed(lens_all, args.patch_dim), dev):
            cohort = DeviceCohort.from_dataset(data, dev, seed=args.seed, augmentation=bool(getattr(args, "augmentation", 1)))
            if not fused and not fstepper.tower_inside:
                # --cache_text 1: every note's frozen-tower embedding once, as a device table the feed launch reads
                with torch.no_grad():
                    cohort.set_text(torch.cat([model.clinic_extractor(cohort.ids[i:i + 32]) for i in range(0, cohort.nb, 32)], 0))
            print(f"cohort resident in HBM: {cohort.nb} bags, {cohort.x.shape[0]} rows x {cohort.F}, "
                  f"{cohort.x.numel() * 4 / 2 ** 30:.2f} GiB")
        else:
            print("cohort does not fit the device next to the working set: host pipeline per step")
    ct_pool = None
    if not fused and fstepper is not None and "CT" in args.modality:
        # synthetic stand-in for the CT encoder's output (aggregator.py:139-140): a small device-resident pool drawn ONCE
        # (it was a 1.3 MB host tensor built inside the training loop every step)
        g_ct = torch.Generator(device=dev).manual_seed(args.seed + 7919)
        ct_pool = torch.randn((16,) + CT_SHAPE, device=dev, generator=g_ct)

    for epoch in range(args.start_epoch, args.n_epochs):
        idx = shard_indices(len(data), world, rank, epoch)                               # sampler.set_epoch(epoch)
        if cohort is not None:
            cohort.draw_epoch(epoch, augment=True)                                       # one launch: this epoch's patch drop
        lr = scheduled_lr(lr0, epoch, args.n_epochs, args.schedule, args.cos)
        losses, accs, bt = AverageMeter("Loss", ":.4e"), AverageMeter("Acc", ":6.3f"), AverageMeter("Time", ":6.3f")
        steps = min(args.iter_per_epoch, len(idx) // per_gpu)
        progress = ProgressMeter(steps, [bt, losses, accs], prefix=f"Epoch: [{epoch}]")
        model.train()
        end = time.time()
        for it in range(steps):
            take = idx[it * per_gpu:(it + 1) * per_gpu]
            if cohort is not None:
                ks = cohort.lengths(take)
                if fused:
                    tr.lr = lr
                    slot = stepper.slot(sum(ks))
                    cohort.feed(take, slot.x, slot.layout.bag_len_dev, slot.y)
                    loss, prob = stepper.step(slot, ks, on_device=True)
                    y, nb_ = slot.y, len(take)
                else:
                    for g in optimizer.param_groups:
                        g["lr"] = lr
                    slot = fstepper.slot(sum(ks))
                    if slot.bucket.fits(ks):
                        if fstepper.tower_inside:
                            cohort.feed(take, slot.x, slot.bucket.len_dev, slot.y, ids_dst=slot.ids)
                        else:
                            cohort.feed(take, slot.x, slot.bucket.len_dev, slot.y, text_dst=slot.text)
                        if slot.ct is not None:
                            for b_, j_ in enumerate(take):
                                slot.ct[b_].copy_(ct_pool[int(j_) % ct_pool.shape[0]], non_blocking=True)
                        loss, prob, _ = fstepper.step(slot, ks, on_device=True)
                        y, nb_ = slot.y, len(take)
                    else:
                        slot = None
                if slot is not None:
                    if it % 10 == 0 or it == steps - 1:
                        losses.update(float(loss.detach()), nb_)
                        accs.update(float(calculate_accuracy(prob.detach(), y)), nb_)
                        bt.update(time.time() - end)
                        progress.display(it)
                    end = time.time()
                    continue
            batch = collate_bags([data[j] for j in take])
            x = batch["pathology"].to(dev, non_blocking=True)
            y = batch["label"].to(dev, non_blocking=True)
            if fused:
                tr.lr = lr
                if stepper is not None and len(batch["lengths"]) == per_gpu:
                    slot = stepper.slot(sum(batch["lengths"]))
                    r0 = 0
                    for b, n in enumerate(batch["lengths"]):                              # the bucket's static input buffers
                        slot.x[r0:r0 + n].copy_(x[b, :n], non_blocking=True)
                        r0 += n
                    slot.y.copy_(y, non_blocking=True)
                    loss, prob = stepper.step(slot, batch["lengths"])
                else:
                    flat = torch.cat([x[b, :n] for b, n in enumerate(batch["lengths"])], 0)
                    loss, prob = tr.train_step(flat, BagLayout.make(batch["lengths"], dev), y)
            else:
                for g in optimizer.param_groups:
                    g["lr"] = lr
                lengths = batch["lengths"]
                slot = None
                if fstepper is not None and len(lengths) == per_gpu:
                    slot = fstepper.slot(sum(lengths))
                    if not slot.bucket.fits(lengths):        # a bag too short for the bucket's kernels: exact-shape step
                        slot = None
                if slot is not None:
                    r0 = 0
                    for b, n in enumerate(lengths):                                       # the bucket's static input buffers
                        slot.x[r0:r0 + n].copy_(x[b, :n], non_blocking=True)
                        r0 += n
                    slot.y.copy_(y, non_blocking=True)
                    if slot.ct is not None:
                        for b_, j_ in enumerate(take):
                            slot.ct[b_].copy_(ct_pool[int(j_) % ct_pool.shape[0]], non_blocking=True)
                    if fstepper.tower_inside:
                        slot.ids.copy_(batch["CI"], non_blocking=True)                    # the text tower runs inside the step
                    else:
                        fstepper.encode_notes(slot, batch["CI"].to(dev))                  # --cache_text 1: a lookup per note
                    loss, prob, _ = fstepper.step(slot, lengths)
                elif graphed is not None:
                    key = (tuple(int(v) for v in lengths), args.variant)
                    if args.variant == "image_only":
                        def body(x_, y_):
                            prob_ = generator([x_], lengths)[1]
                            return criterion(prob_, y_), prob_
                        loss, prob = graphed.run(key, (x, y), body)
                    elif args.learnablePrompt:
                        def body(x_, ids_, y_):
                            prob_ = generator([x_], ids_, lengths, labels=y_ if fuse_loss else None)[0]
                            return loss_of(prob_, y_), prob_
                        loss, prob = graphed.run(key, (x, batch["CI"].to(dev), y), body)
                    else:
                        tfeat = model.clinic_extractor(batch["CI"].to(dev))      # frozen tower (no_grad): outside the graph
                        def body(x_, t_, y_):
                            prob_ = generator([x_], None, lengths, text_features=t_, labels=y_ if fuse_loss else None)[0]
                            return loss_of(prob_, y_), prob_
                        loss, prob = graphed.run(key, (x, tfeat, y), body)
                    optimizer.step()
                else:
                    if args.variant == "image_only":
                        _, prob = generator([x], lengths)
                        loss = criterion(prob, y)
                    else:
                        xs = [x]
                        if "CT" in args.modality:
                            # the CT encoders are outside the hot path: their OUTPUT (the feature map of aggregator.py:139-140)
                            # is an input here - synthetic, like the bags
                            from . import synthetic as syn
                            ct = syn.make_ct_map(args.seed + 7919 * epoch + it, x.shape[0], 160, 2).to(dev)
                            xs = [ct, x] if "pathology" in args.modality else [ct]
                        ls = 3.0 / (x.shape[0] * (1 if args.num_classes > 2 else args.num_classes)) if three_terms else None
                        out = generator(xs, batch["CI"].to(dev), lengths, labels=y if fuse_loss else None, loss_scale=ls)
                        prob, toks = unpack(out)
                        loss = total_loss(prob, toks, y)                                  # loss_point 'Last' by default
                    optimizer.zero_grad()
                    ops.backward(loss)
                    optimizer.step()
            if it % 10 == 0 or it == steps - 1:
                losses.update(float(loss.detach()), x.shape[0])                                   # host sync only when logging
                accs.update(float(calculate_accuracy(prob.detach(), y)), x.shape[0])
                bt.update(time.time() - end)
                progress.display(it)
            end = time.time()
        if rank == 0 and args.save_dir:
            os.makedirs(args.save_dir, exist_ok=True)
            # same key schema either way (the model's own names), optimizer state included: a fused-step checkpoint
            # loads strictly into the model / test_ddp.py and resumes like the autograd route's
            sd = tr.model_state_dict() if fused else model.state_dict()
            state = {"epoch": epoch + 1, "state_dict": sd,
                     "optimizer": tr.optimizer_state_dict() if fused else optimizer.state_dict()}
            save_checkpoint(state, True, args.save_dir, f"checkpoint_{epoch:04d}.pth.tar")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args = create_arg_parser(argv)
    if not torch.cuda.is_available():
        raise NotImplementedError("the MIL hot path runs on MI355X only: no GPU is visible")   # train_ddp.py:83-88
    if args.multiprocessing_distributed:
        n = len(args.gpu.split(","))
        mp.spawn(main_worker, nprocs=n, args=(n, args))
    else:
        main_worker(0, 1, args)


if __name__ == "__main__":
    main()
