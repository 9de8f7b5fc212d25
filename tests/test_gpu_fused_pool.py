"""GPU: the attention-pool partial pass fused into the gate forward's epilogue (k_gate_fwd2<.., PQ = 2>, what the one-call
step runs when every tile of the batch is a full 32-row tile) against the stand-alone k_pool_partial launch
(MIL_FUSE_POOL=0): same partial sums, head projections, logits, loss and gradients - in eval mode and in model.train() mode
(same Philox stream: the fused kernel re-draws the head mask words workgroup 0 writes out)."""
import os

import pytest
import torch

from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer

pytestmark = pytest.mark.gpu


def _run(train_mode, fuse, B, N, L=512, bf16=False):
    dev = torch.device("cuda")
    old = os.environ.get("MIL_FUSE_POOL")
    os.environ["MIL_FUSE_POOL"] = "1" if fuse else "0"
    os.environ["MIL_FUSE_POOL16"] = "1" if fuse else "0"        # bf16: the fused form is the default in eval mode only
    try:
        p = syn.image_only_params(1234, L=L)
        tr = ImageOnlyTrainer(p, dev, train_mode=train_mode)
        x = syn.make_bags(4321, B, N, L).reshape(B * N, L).to(dev)
        if bf16:
            x = x.to(torch.bfloat16)
        y = syn.make_labels(99, B).to(dev)
        lay = BagLayout.uniform(B, N, dev)
        tr.forward(x, lay, y)
        tr.backward()
        torch.cuda.synchronize()
        T = lay.T
        out = dict(partials=tr._keep[3]["partials"][:T * (L + 2)].clone(), hrow=tr.last["hrow"].clone(),
                   logits=tr.last["logits"].clone(), ds=tr.last["ds"].clone(), grad=tr.fp.grad.clone(),
                   loss=tr.loss_sum.clone(), scores=tr.last["scores"].clone())
        if train_mode:
            out["mbits"] = tr.last["mbits"].clone()
            out["xbits"] = tr.last["xbits"].clone()
        return out
    finally:
        os.environ.pop("MIL_FUSE_POOL16", None)
        if old is None:
            os.environ.pop("MIL_FUSE_POOL", None)
        else:
            os.environ["MIL_FUSE_POOL"] = old


@pytest.mark.parametrize("train_mode", [False, True])
@pytest.mark.parametrize("B,N", [(32, 1024), (40, 832)])       # 40 x 832 = 260 workgroups: the last one has two live tiles
def test_fused_pool_pass_equals_the_stand_alone_launch(train_mode, B, N):
    a = _run(train_mode, True, B, N)
    b = _run(train_mode, False, B, N)
    assert torch.equal(a["scores"], b["scores"])
    if train_mode:
        assert torch.equal(a["xbits"], b["xbits"]) and torch.equal(a["mbits"], b["mbits"])
    # same arithmetic, operation for operation; the compiler may contract the two kernels' dot products differently, so the
    # comparison allows the last bits
    for k in ("partials", "hrow", "logits", "ds", "grad", "loss"):
        d = float((a[k] - b[k]).abs().max())
        ref = float(b[k].abs().max())
        assert d <= 2e-6 * max(1.0, ref), (k, d, ref)
    assert bool(torch.isfinite(a["grad"]).all())


def _run_tail(train_mode, short, B, N):
    old = os.environ.get("MIL_TAIL_H")
    os.environ["MIL_TAIL_H"] = "1" if short else "0"
    try:
        return _run(train_mode, True, B, N)
    finally:
        if old is None:
            os.environ.pop("MIL_TAIL_H", None)
        else:
            os.environ["MIL_TAIL_H"] = old


@pytest.mark.parametrize("train_mode", [False, True])
@pytest.mark.parametrize("B,N", [(32, 1024), (40, 832), (3, 5000)])      # 5000 rows: more tiles than the register passes hold
def test_short_chain_tail_equals_the_merge_first_tail(train_mode, B, N):
    """k_pool_tail_h (logits from the rows' head projections, M merged behind the ds stores) against k_pool_merge_head (M
    first, logits from M): the same numbers up to summation order."""
    a = _run_tail(train_mode, True, B, N)
    b = _run_tail(train_mode, False, B, N)
    assert torch.equal(a["scores"], b["scores"]) and torch.equal(a["partials"], b["partials"])
    for k in ("logits", "ds", "grad", "loss"):
        d = float((a[k] - b[k]).abs().max())
        ref = float(b[k].abs().max())
        assert d <= 3e-6 * max(1.0, ref), (k, d, ref)
    assert bool(torch.isfinite(a["grad"]).all())


@pytest.mark.parametrize("train_mode", [False, True])
@pytest.mark.parametrize("B,N", [(3, 5000), (1, 10016), (2, 4096), (1, 40000)])     # 40 000 rows: more tiles than 16 x 4 passes
def test_long_bag_tail_in_two_launches_equals_the_one_launch_tail(train_mode, B, N):
    """k_tail_stats + k_tail_apply (>= 64 tiles per bag on average: many workgroups per bag) against k_pool_tail_h (MIL_TAIL_H=1:
    one launch, two workgroups per bag): the same numbers up to summation order."""
    old = os.environ.pop("MIL_TAIL_H", None)
    try:
        a = _run(train_mode, True, B, N)
    finally:
        if old is not None:
            os.environ["MIL_TAIL_H"] = old
    b = _run_tail(train_mode, True, B, N)
    assert torch.equal(a["scores"], b["scores"]) and torch.equal(a["partials"], b["partials"])
    for k in ("logits", "ds", "grad", "loss"):
        d = float((a[k] - b[k]).abs().max())
        ref = float(b[k].abs().max())
        assert d <= 3e-6 * max(1.0, ref), (k, d, ref)
    assert bool(torch.isfinite(a["grad"]).all())


@pytest.mark.parametrize("train_mode", [False, True])
@pytest.mark.parametrize("B,N,L", [(32, 4096, 1024), (17, 4096, 512), (33, 2016, 1024)])
def test_bf16_fused_pool_pass_equals_the_stand_alone_launch(train_mode, B, N, L):
    """Round 4: the bf16 deep forward (k_gate_fwd_bf16_deep<2, .., PQ>) with the pool partial pass in its epilogue - one wave
    per 32-row tile playing k_pool_partial_bf16's four waves in turn - against the stand-alone pass (MIL_FUSE_POOL=0).
    32 x 4096 x 1024 is BASELINE config 5; 17 x 4096 x 512 the one-block form; 33 x 2016 = 66 528 rows = 259.9 workgroups of
    256 rows: the last one has seven live tiles."""
    a = _run(train_mode, True, B, N, L, bf16=True)
    b = _run(train_mode, False, B, N, L, bf16=True)
    assert torch.equal(a["scores"], b["scores"])
    if train_mode:
        assert torch.equal(a["xbits"], b["xbits"]) and torch.equal(a["mbits"], b["mbits"])
    for k in ("partials", "hrow", "logits", "ds", "grad", "loss"):
        d = float((a[k] - b[k]).abs().max())
        ref = float(b[k].abs().max())
        assert d <= 2e-6 * max(1.0, ref), (k, d, ref)
    assert bool(torch.isfinite(a["grad"]).all())
