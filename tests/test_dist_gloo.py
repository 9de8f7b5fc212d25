"""CPU, world_size 2, gloo: the data-parallel step's exchange logic.  Each rank computes the gradients of ITS
shard of bags (the oracle stands in for the HIP kernels, which need a GPU) with the loss normalised by the
GLOBAL bag count, packs them in the flat buffer and runs the trainer's single all-reduce; the result must
equal the single-process gradient of the whole batch, and parameters must be rank 0's after the broadcast."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mil_amd import synthetic as syn
from mil_amd.dist_utils import allreduce_flat, broadcast_flat, init_process_group, shard_indices
from mil_amd.trainer import FlatParams, PARAM_ORDER
from oracle import mil_oracle as orc

N_BAGS, N, L = 6, 48, 512


def _bag(i):
    return torch.randn((N, L), generator=torch.Generator().manual_seed(900 + i))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    init_process_group("gloo", "env://", world, rank)
    torch.set_num_threads(2)
    # rank-dependent init, then the DDP-style broadcast of rank 0's parameters
    p = syn.image_only_params(1234 + 17 * rank, L=L)
    fp = FlatParams(p, "cpu", PARAM_ORDER)
    broadcast_flat(fp.flat, 0)
    params = {k: fp.p(k).clone() for k in PARAM_ORDER}
    labels = syn.make_labels(5, N_BAGS)
    mine = shard_indices(N_BAGS, world, rank, epoch=0)
    leaves = {k: params[k].clone().requires_grad_(True) for k in PARAM_ORDER}
    outs = [orc.image_only_forward(_bag(i), leaves) for i in mine]
    prob = torch.cat([o["prob"] for o in outs], 0)
    y = labels[mine]
    lp = torch.clamp(torch.log(prob), min=-100.0)
    l1p = torch.clamp(torch.log(1 - prob), min=-100.0)
    loss = (-(y * lp + (1 - y) * l1p)).sum() / (N_BAGS * 2)          # scale = 1 / (C * GLOBAL bags)
    loss.backward()
    for k in PARAM_ORDER:
        fp.g(k).copy_(leaves[k].grad)
    allreduce_flat(fp.grad)                                           # the step's one collective
    torch.save({"flat": fp.flat.clone(), "grad": fp.grad.clone(), "mine": mine}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_allreduce_equals_full_batch(tmp_path):
    world, port = 2, 29600 + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert torch.equal(r0["flat"], r1["flat"])                        # broadcast: everyone holds rank 0's weights
    assert torch.equal(r0["grad"], r1["grad"])
    assert sorted(r0["mine"] + r1["mine"]) == list(range(N_BAGS))
    p = syn.image_only_params(1234, L=L)
    ref = FlatParams(p, "cpu", PARAM_ORDER)
    assert torch.equal(ref.flat, r0["flat"])
    loss, _, _, grads = orc.batch_loss_and_grads([_bag(i) for i in range(N_BAGS)], syn.make_labels(5, N_BAGS), p)
    for k in PARAM_ORDER:
        o = ref.offsets[k]
        got = r0["grad"][o:o + grads[k].numel()].view(grads[k].shape)
        assert float((got - grads[k]).abs().max()) <= 1e-6 * max(1.0, float(grads[k].abs().max())), k
