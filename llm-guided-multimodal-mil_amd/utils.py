"""Training-loop helpers (reference: utils.py:120-241; the cohort / DICOM helpers are out of scope)."""
import math
import os
import shutil

import torch


class AverageMeter:
    def __init__(self, name: str, fmt: str = ":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.sum = self.avg = 0.0
        self.count = 0

    def update(self, val, n: int = 1):
        self.val = float(val)
        self.sum += float(val) * n
        self.count += n
        self.avg = self.sum / max(1, self.count)

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


class ProgressMeter:
    def __init__(self, num_batches: int, meters, prefix: str = ""):
        self.width = len(str(num_batches))
        self.total, self.meters, self.prefix = num_batches, meters, prefix

    def display(self, batch: int):
        head = f"{self.prefix}[{batch:{self.width}d}/{self.total}]"
        print("\t".join([head] + [str(m) for m in self.meters]), flush=True)


def calculate_accuracy(outputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """Fraction of bags whose top-1 class equals the one-hot label's class (reference utils.py:159-171)."""
    with torch.no_grad():
        return (outputs.argmax(dim=1) == targets.argmax(dim=1)).float().mean()


def save_checkpoint(state: dict, is_best: bool, save_dir: str, filename: str = "checkpoint.pth.tar"):
    path = os.path.join(save_dir, filename)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(save_dir, "checkpoint_best.pth.tar"))


def scheduled_lr(base_lr: float, epoch: int, n_epochs: int, schedule, cos: bool) -> float:
    """Cosine or step (x0.1 at each milestone) schedule, as utils.py:232-241."""
    if cos:
        return base_lr * 0.5 * (1.0 + math.cos(math.pi * epoch / n_epochs))
    lr = base_lr
    for milestone in schedule:
        if epoch >= milestone:
            lr *= 0.1
    return lr


def adjust_learning_rate(optimizer, epoch: int, args):
    lr = scheduled_lr(args.lr, epoch, args.n_epochs, args.schedule, args.cos)
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


class CLIPloss_v1(torch.nn.Module):
    """CLIP-as-loss of the image-only variant (reference utils.py:247-284): contrast every bag embedding with the
    frozen CLIP text features of the per-feature prompts of every sample in the batch.

    The reference builds the prompts as strings ("a lung cancer patient photo of <feature> <value>", :266-267) and
    tokenises them; text handling is out of scope here, so forward takes the token ids int64 [b, F, 77] directly.
    `clip.load` downloads weights: the tower is the ViT-B/32 text architecture with caller-loaded or random weights."""

    def __init__(self, args, text_model=None):
        super().__init__()
        from .clip.model import CLIPText
        self.args = args
        self.model = text_model if text_model is not None else CLIPText(
            512, 77, int(getattr(args, "clip_vocab", 49408)), int(getattr(args, "clip_width", 512)),
            int(getattr(args, "clip_heads", 8)), int(getattr(args, "clip_layers", 12)))
        for p in self.model.parameters():
            p.requires_grad_(False)

    def forward(self, output: torch.Tensor, prompt_ids: torch.Tensor) -> torch.Tensor:
        from . import ops
        b, F_, ctx = prompt_ids.shape
        with torch.no_grad():
            feat = self.model.encode_text(prompt_ids.reshape(b * F_, ctx)).reshape(b, F_, -1)
        return ops.clip_contrastive_loss(output, feat)
