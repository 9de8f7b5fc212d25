"""GPU: bench.py's own launcher.  `python bench.py --gpus N` without torchrun must start N ranks itself (reference:
mp.spawn of one process per GPU, train_ddp.py:53-82,622-624), never fall back to one rank, and the N>1 step must equal
the single-process step on the union of the ranks' bags."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import rel_err
from mil_amd import synthetic as syn
from mil_amd.bags import BagLayout
from mil_amd.trainer import ImageOnlyTrainer

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--prime", "2", "--bags-per-gpu", "4", "--patches", "128", "--no-cpu-baseline",
         "--no-configs", "--no-breakdown"]


def bench(*argv, env=None, expect_ok=True):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=e)
    if expect_ok:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    return r


def test_one_rank_through_rccl_reports_the_collective():
    line = bench("--gpus", "1", *SMALL, env={"MIL_FORCE_COLLECTIVES": "1"})
    assert line["n_gpus"] == 1 and line["rccl"]["backend"] == "nccl" and line["rccl"]["world_size"] == 1
    assert line["rccl"]["allreduce_bytes"] > 700000          # the flat gradient + loss slot (0.79 MB)
    assert line["rccl"]["allreduce_us"] > 0


@pytest.mark.parametrize("W", [2, 4])
def test_ranks_rehearsal_equals_the_single_process_step(tmp_path, W):
    """W ranks started by bench.py itself, all on GPU 0 over gloo (rehearsal; 4 ranks stay inside the box's limit of 6
    processes on the card): the all-reduced gradient and loss equal ONE process on the union of the ranks' bags."""
    dump = str(tmp_path / "g.pt")
    line = bench("--gpus", str(W), *SMALL, "--train-mode", "0", "--dump", dump, env={"MIL_BENCH_REHEARSAL": "1"})
    assert line["n_gpus"] == W and line["config"]["global_bags"] == 4 * W
    assert line["rccl"]["world_size"] == W and line["rccl"]["backend"] == "gloo"          # rehearsal: every rank on GPU 0
    got = torch.load(dump, weights_only=True)
    # the same bags (rank r draws make_bags(4321 + r) / make_labels(99 + r)) in ONE process
    dev = torch.device("cuda")
    B, N, L = 4, 128, 512
    x = torch.cat([syn.make_bags(4321 + r, B, N, L).reshape(B * N, L) for r in range(W)], 0).to(dev)
    y = torch.cat([syn.make_labels(99 + r, B, 2) for r in range(W)], 0).to(dev)
    tr = ImageOnlyTrainer(syn.image_only_params(1234, L=L), dev)
    tr.forward(x, BagLayout.uniform(W * B, N, dev), y)
    tr.backward()
    assert abs(float(tr.loss_sum.item()) - got["loss"]) <= 1e-6
    assert rel_err(got["grad"], tr.fp.grad.cpu()) <= 1e-5


def test_strong_scaling_rehearsal_keeps_the_global_batch(tmp_path):
    """`--global-bags G` (config 4 for W < 8 keeps the 256-bag global batch, 256 / W per GPU): two ranks of G / 2
    bags each reproduce the one-process gradient on the same G bags; the line says `"scaling": "strong"` and lists every
    timed region."""
    dump = str(tmp_path / "g.pt")
    small = [a_ for a_ in SMALL if a_ not in ("--bags-per-gpu", "4")]
    line = bench("--gpus", "2", *small, "--global-bags", "8", "--regions", "3", "--train-mode", "0", "--dump", dump,
                 env={"MIL_BENCH_REHEARSAL": "1"})
    assert line["scaling"] == "strong" and line["config"]["global_bags"] == 8 and line["config"]["bags_per_gpu"] == 4
    assert len(line["ms_per_step_runs"]) == 3
    assert line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    assert line["ms_per_step"] == sorted(line["ms_per_step_runs"])[1]                    # the median region
    got = torch.load(dump, weights_only=True)
    dev = torch.device("cuda")
    B, N, L = 4, 128, 512
    x = torch.cat([syn.make_bags(4321 + r, B, N, L).reshape(B * N, L) for r in range(2)], 0).to(dev)
    y = torch.cat([syn.make_labels(99 + r, B, 2) for r in range(2)], 0).to(dev)
    tr = ImageOnlyTrainer(syn.image_only_params(1234, L=L), dev)
    tr.forward(x, BagLayout.uniform(8, N, dev), y)
    tr.backward()
    assert abs(float(tr.loss_sum.item()) - got["loss"]) <= 1e-6
    assert rel_err(got["grad"], tr.fp.grad.cpu()) <= 1e-5
    r = bench("--gpus", "2", *small, "--global-bags", "7", env={"MIL_BENCH_REHEARSAL": "1"}, expect_ok=False)
    assert r.returncode != 0 and "multiple" in (r.stderr + r.stdout)


def test_more_ranks_than_gpus_is_refused_not_downgraded():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    r = bench("--gpus", "2", *SMALL, expect_ok=False)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not r.stdout.strip()                       # no JSON line with a wrong n_gpus


def test_gpus_flag_must_match_torchrun_world_size():
    r = bench("--gpus", "2", *SMALL, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_ok=False)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_dying_rank_takes_the_job_down():
    """One of two ranks exits non-zero between warm-up and the timed region while the other waits in the barrier
    (reference: mp.spawn joins and re-raises, train_ddp.py:622-624): bench.py must return non-zero well inside its
    timeout and leave no rank behind."""
    import time
    import psutil
    marker = "777.25"                                   # a unique --launch-timeout value to find the job's processes by
    t0 = time.time()
    r = bench("--gpus", "2", *SMALL, "--launch-timeout", marker, env={"MIL_BENCH_REHEARSAL": "1", "MIL_BENCH_FAIL_RANK": "1"},
              expect_ok=False)
    took = time.time() - t0
    assert r.returncode != 0 and "rank 1 exited with 3" in r.stderr, r.stderr[-2000:]
    assert took < 240, took
    assert not r.stdout.strip().startswith("{")          # no result line from a broken job
    time.sleep(1.0)
    left = [p.pid for p in psutil.process_iter(["cmdline"]) if p.info["cmdline"] and marker in p.info["cmdline"]]
    assert left == [], left
