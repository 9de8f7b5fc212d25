"""CPU: host-side logic of the drop-in (no kernels): sharding, schedules, flat parameter views, collation,
segment maps, flag parsing."""
import numpy as np
import torch
from torch.utils.data import DistributedSampler

from mil_amd.config import create_arg_parser
from mil_amd.dataset import SyntheticBags, collate_bags
from mil_amd.dist_utils import shard_indices
from mil_amd.segments import AttnSegs
from mil_amd.trainer import FlatParams, PARAM_ORDER
from mil_amd.utils import AverageMeter, calculate_accuracy, scheduled_lr
from mil_amd import synthetic as syn


class _DS:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def test_shard_indices_equal_torch_distributed_sampler():
    for n in (37, 64, 5):
        for world in (1, 2, 4, 8):
            for epoch in (0, 1, 7):
                seen = []
                for rank in range(world):
                    s = DistributedSampler(_DS(n), num_replicas=world, rank=rank, shuffle=True)
                    s.set_epoch(epoch)
                    mine = shard_indices(n, world, rank, epoch)
                    assert mine == list(s)
                    seen += mine
                assert set(seen) == set(range(n))


def test_lr_schedule():
    assert scheduled_lr(1e-5, 0, 1000, [500], False) == 1e-5
    assert abs(scheduled_lr(1e-5, 500, 1000, [500], False) - 1e-6) < 1e-18
    assert abs(scheduled_lr(1e-5, 500, 1000, [500], True) - 0.5e-5) < 1e-12


def test_flat_params_views_alias_one_buffer():
    p = syn.image_only_params(1, L=512)
    fp = FlatParams(p, "cpu", PARAM_ORDER)
    assert fp.numel >= sum(v.numel() for v in p.values())
    for k in PARAM_ORDER:
        assert torch.equal(fp.p(k), p[k])
        assert fp.offsets[k] % 4 == 0                          # 16-byte aligned views for the float4 kernels
    fp.p("fc.1.bias").add_(1.0)
    o = fp.offsets["fc.1.bias"]
    assert torch.equal(fp.flat[o:o + 2], p["fc.1.bias"] + 1.0)
    assert set(fp.state_dict().keys()) == set(PARAM_ORDER)


def test_collate_pads_and_keeps_lengths():
    ds = SyntheticBags(6, 40, 16, prompts=1, ragged=True, seed=3)
    b = collate_bags([ds[i] for i in range(4)])
    assert b["pathology"].shape[0] == 4 and b["pathology"].shape[1] == max(b["lengths"])
    for i, n in enumerate(b["lengths"]):
        assert float(b["pathology"][i, n:].abs().sum()) == 0.0
        assert torch.equal(b["pathology"][i, :n], ds[i]["pathology"])
    assert b["CI"].dtype == torch.int64 and tuple(b["CI"].shape) == (4, 1, 77)
    assert (b["CI"].argmax(-1) > 0).all()                     # EOT is the row maximum


def test_attention_segment_maps():
    s = AttnSegs([1, 10], [70, 129], "cpu")
    assert s.q_off.tolist() == [0, 1, 11] and s.k_off.tolist() == [0, 70, 199]
    assert s.bag_tile_off.tolist() == [0, 2, 5]
    tm = s.tile_map.numpy()
    cover = np.zeros(199, int)
    for bag, k0, n in tm:
        cover[k0:k0 + n] += 1
        assert n <= 64
    assert (cover == 1).all()
    assert s.q_bag.tolist() == [0] + [1] * 10


def test_flags_and_meters():
    a = create_arg_parser(["--modality", "['pathology']", "--synthetic", "[128, 768, 16]", "--batch_size", "4"])
    assert a.modality == ["pathology"] and a.synthetic == [128, 768, 16] and a.aggregator == "ABMIL"
    m = AverageMeter("Loss", ":.3f")
    m.update(1.0, 2); m.update(4.0, 1)
    assert abs(m.avg - 2.0) < 1e-12
    out = torch.tensor([[0.4, 0.6], [0.7, 0.3]]); tgt = torch.tensor([[0.0, 1.0], [0.0, 1.0]])
    assert float(calculate_accuracy(out, tgt)) == 0.5


def test_two_segment_layout_covers_patch_and_token_rows():
    """[all patches | all tokens] row order: every row belongs to exactly one tile of the right bag."""
    import torch
    from mil_amd.bags import BagLayout
    n, t = [70, 5, 33], [1, 1, 1]
    lay = BagLayout.two_segment(n, t, torch.device("cpu"))
    tm, bto = lay.tile_map.numpy(), lay.bag_tile_off.numpy()
    assert lay.R == sum(n) + sum(t) and lay.B == 3 and lay.T == tm.shape[0] == bto[-1]
    owner = -np.ones(lay.R, dtype=np.int64)
    for b in range(3):
        for bag, row0, nrows, _ in tm[bto[b]:bto[b + 1]]:
            assert bag == b and 0 < nrows <= 32
            assert (owner[row0:row0 + nrows] == -1).all()
            owner[row0:row0 + nrows] = b
    koff = np.concatenate([[0], np.cumsum(n)])
    for b in range(3):
        assert (owner[koff[b]:koff[b + 1]] == b).all()
        assert owner[sum(n) + b] == b


def test_aligned32_marks_only_plain_layouts_of_whole_tiles():
    """MIL_STAGE_POOL_FUSED may only be set for a single-segment map whose tile t covers rows 32 t .. 32 t + 31
    (include/mil_hip.h): BagLayout.aligned32 is what ImageOnlyTrainer derives the hint from."""
    from mil_amd.bags import BagLayout
    dev = torch.device("cpu")
    a = BagLayout.make([64, 96, 32], dev)
    assert a.aligned32 and a.T * 32 == a.R
    tm = a.tile_map.numpy()
    assert (tm[:, 1] == 32 * np.arange(a.T)).all() and (tm[:, 2] == 32).all()
    assert not BagLayout.make([64, 40], dev).aligned32                      # a bag ending in a partial tile
    two = BagLayout.two_segment([64, 64], [32, 32], dev)                   # all lengths multiples of 32, but tiles bag-major
    assert not two.aligned32 and two.T * 32 == two.R
    assert (two.tile_map.numpy()[:, 1] != 32 * np.arange(two.T)).any()
