#!/usr/bin/env python3
"""bags/sec, forward+backward(+all-reduce+Adam), N=1024 patches, D=512 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # starts N ranks itself when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic bags resident in HBM:
per GPU 32 bags x 1024 patches x 512 dims (BASELINE config 2 at N=1; 8 GPUs x 32 = the 256-bag
config 4), image-only branch: gate scores (fp32 MFMA) -> attention pool -> head -> BCE ->
backward -> one flat-gradient RCCL all-reduce -> Adam.  Weak scaling: per-GPU work is fixed.

Launching (reference: mp.spawn of one process per GPU, train_ddp.py:53-82,622-624).  With WORLD_SIZE in the
environment this process IS one rank (torchrun).  Without it and with --gpus N > 1 this process starts N child
ranks BEFORE touching a GPU (fresh python processes, rank r bound to GPU r, backend nccl = RCCL), waits for
them and exits non-zero if any rank fails or fewer than N devices are visible: it never falls back to one rank.

Rank 0 prints ONE JSON line.  Extra objects on it:
  roofline      dominant kernel AMONG THE LAUNCHES OF THE STEP (fp32 MFMA bound), algorithmic flops / its HIP-event time
                measured inside the running step (an event between the step's launch groups, a second pass right after
                the timed region); profiles/rNN_cfg2_kernel_stats.csv holds rocprofv3's average for the same kernel
  roofline_step the whole driver-timed step: 2 x 4 R L D flop / ms_per_step against the same peak
  roofline_pool the HBM-bound attention-pool kernel at N=4096, D=512 (north_star's 30 % target)
  cpu_baseline  the CPU oracle (torch fp32, one bag per forward as the reference runs) on this host, at 1 thread
                and at all physical cores
  rccl          (N>1 or MIL_FORCE_COLLECTIVES=1) world size RCCL reports, measured all-reduce time of the step's buffer
  configs       (N=1) BASELINE config 3 (CLIP-text fusion, 32 x 1024 x 768) and config 5 (bf16, 32 x 4096 x 1024):
                ms/step, bags/s, dominant-kernel roofline, and a parity check of the SAME full-size batch against
                the oracle on 2 of the 32 bags (outside the timed region)
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16 MFMA, dense
PEAK_HBM_GBS = 8000.0            # HBM3E spec
D_GATE = 192
# HBM bytes per launch from rocprofv3 PMC passes of this same command at the default workload
# (profiles/*_hbm_traffic_pmc.csv: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction applied)
PMC_TRAFFIC_FILE = os.path.join(REPO, "profiles", "pmc_traffic.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--prime", type=int, default=300, help="untimed forward passes before the warm-up steps (clock ramp)")
    ap.add_argument("--batches", type=int, default=4,
                    help="distinct synthetic batches resident in HBM that the steps cycle through (1: the same 64 MiB every "
                         "step, which then sits in the 256 MB Infinity Cache from one step to the next - no loader does that)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--regions", type=int, default=7,
                    help="the timed region (exactly --steps steps between two barrier + synchronize pairs) is run this many times "
                         "back to back; ms_per_step is the MEDIAN region, every region is listed in ms_per_step_runs")
    ap.add_argument("--bags-per-gpu", type=int, default=32)
    ap.add_argument("--global-bags", type=int, default=0,
                    help="strong scaling (SURVEY 8d, config 4 for W < 8): G bags in total, G / W per GPU (G % W == 0); "
                         "default 0 = weak scaling with --bags-per-gpu bags on every GPU")
    ap.add_argument("--patches", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: BASELINE config 5 variant (x and gate weights stored bf16, fp32 accumulate); use with "
                         "--patches 4096 --dim 1024.  The headline metric is f32.")
    ap.add_argument("--train-mode", type=int, default=1, help="1: a true model.train() step - dropout(0.5) on the patches "
                    "(ABMIL.py:49) and dropout(0.25) before the head (aggregator.py:129), masks generated in-kernel; "
                    "0: the eval-mode (parity) step of round 1")
    ap.add_argument("--accum", type=int, default=1, help="micro-batches per optimizer step (gradient accumulation): "
                    "one all-reduce + Adam every ACCUM passes; a 'step' stays one pass over one batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / config 5 / ragged-regime objects")
    ap.add_argument("--no-ragged", action="store_true", help="skip configs.ragged_image / ragged_fusion / ragged_ct_pth "
                    "(one ragged bag per step fed from the HBM-resident cohort)")
    ap.add_argument("--no-rccl-floor", action="store_true", help="skip the rccl_floor object (two child runs at world size 1 with "
                    "the RCCL all-reduce forced into the step)")
    ap.add_argument("--only-ragged", default="", help="(tools) run just this regime object (image | fusion | ct_pth) and print it")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward as one hipGraph (default: eager; "
                    "the step is GPU-bound either way)")
    ap.add_argument("--dump", default="", help="(tests) rank 0 saves loss + flat gradient of the first step here")
    ap.add_argument("--launch-timeout", type=float, default=1500.0)
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args) -> int:
    """Parent of an N-rank run: starts the ranks, never touches a GPU itself (device_count() does not initialise
    HIP on this image).  Returns the exit code for the whole job."""
    import torch
    rehearsal = os.environ.get("MIL_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not rehearsal:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible; refusing to run fewer ranks "
              f"(set MIL_BENCH_REHEARSAL=1 for the one-GPU gloo rehearsal of the N>1 code path)", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + args.launch_timeout
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.1)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
        if (rc != 0 or time.time() > deadline) and live:
            if rc == 0:
                rc = 124
                print("bench.py: launch timeout; stopping the ranks", file=sys.stderr)
            for p in live:                      # exactly the processes started above, by handle
                p.terminate()
            t_end = time.time() + 10
            for p in live:
                try:
                    p.wait(max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            live = []
    return rc


# ----------------------------------------------------------------------------------------------- helpers
def timed(fn, iters, warm=2):
    """Average duration (ms) of fn() on the current stream, measured with HIP events."""
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def _pmc_traffic(name):
    try:
        return json.load(open(PMC_TRAFFIC_FILE)).get(name)
    except (OSError, ValueError):
        return None


def kernel_breakdown(tr, x, lay, y, iters=20):
    """HIP-event time of each launch group of one step (separate pass, outside the timed region).  Each group is timed
    through the library's stand-alone C entry points in one call per launch (tr.time_pieces)."""
    return tr.time_pieces(x, lay, y, iters)


def pool_roofline(dev, iters=20):
    """The HBM-bound attention-pool stage at the north_star point N=4096, D=512: one pass reads x once
    (N*L*4 B), the scores (4 N) and writes M (4 L).  Enough bags to exceed the 256 MiB Infinity Cache
    so the bytes really come from HBM."""
    import torch
    from mil_amd import ops
    from mil_amd.bags import BagLayout
    N, L, B = 4096, 512, 64            # 512 MiB of x
    x = torch.randn(B * N, L, device=dev)
    scores = torch.randn(B * N, device=dev)
    lay = BagLayout.uniform(B, N, dev)
    ms = timed(lambda: ops.attn_pool_fwd(x, scores, lay), iters, warm=3)
    alg_bytes = B * (N * L * 4 + 4 * N + 4 * L)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_pool_partial+k_pool_merge", "workload": f"{B} bags x {N} x {L} fp32",
            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4),
            "traffic": _pmc_traffic("pool_4096x512"), "bytes_per_launch": alg_bytes,
            "ms_per_launch": round(ms, 4), "passes_per_s": round(B / (ms * 1e-3), 1)}


def _physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        n = len(seen)
    except OSError:
        n = 0
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota): threads beyond it only time-slice
    quota = avail
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // per)
        except (OSError, ValueError):
            pass
    return max(1, min(n or avail, avail, quota))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(N, L, budget_s=8.0):
    """The reference's CPU arithmetic (oracle restatement) timed on this host: one bag per forward,
    fp32, eval, fwd + BCE + bwd, at 1 thread and at all physical cores this process may use."""
    import torch
    from mil_amd import synthetic as syn
    from oracle import mil_oracle as orc
    p = syn.image_only_params(1234, L=L)
    names = list(p.keys())
    y = syn.make_labels(1, 1)
    bags = [torch.randn(N, L) for _ in range(4)]

    def one(xb):
        leaves = {k: p[k].clone().requires_grad_(True) for k in names}
        o = orc.image_only_forward(xb, leaves)
        loss = orc.bce_loss(o["prob"], y)
        loss.backward()

    def run(threads):
        torch.set_num_threads(threads)
        for xb in bags[:3]:
            one(xb)
        n, t0 = 0, time.perf_counter()
        while True:
            one(bags[n % 4])
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s:
                return n, el

    cores = _physical_cores()
    n1, e1 = run(1)
    # "all cores": the physical cores this process may use; torch's intra-op pool stops scaling on a 1024 x 512 bag long
    # before 64+ threads, so 16 threads (the best count measured on the EPYC 9575F host) is tried too and the faster kept
    best = None
    for c in sorted({cores, min(cores, 16)}):
        nc_, ec_ = run(c)
        if best is None or nc_ / ec_ > best[1] / best[2]:
            best = (c, nc_, ec_)
    cores_used, nc, ec = best
    return {"value": round(nc / ec, 2), "unit": "bags/s", "cores": cores_used, "cores_available": cores, "kind": "port",
            "sample": f"{nc} bags of {N}x{L} fp32, one bag per fwd+loss+bwd, torch-CPU oracle, {ec:.1f} s",
            "one_thread": {"value": round(n1 / e1, 2), "unit": "bags/s", "cores": 1,
                           "sample": f"{n1} bags in {e1:.1f} s"},
            "cpu": _cpu_model()}


# ----------------------------------------------------------------------------------------------- extra configs (N=1)
def config5_bf16(dev, steps=60, warmup=5):
    """BASELINE config 5: 32 bags x 4096 patches x 1024 dims, x and gate weights stored bf16, fp32 accumulate."""
    import torch
    from mil_amd import ops, synthetic as syn
    from mil_amd.bags import BagLayout
    from mil_amd.trainer import ImageOnlyTrainer
    from oracle import mil_oracle as orc
    B, N, L = 32, 4096, 1024
    p = syn.image_only_params(1234, L=L)
    tr = ImageOnlyTrainer(p, dev)
    x32 = syn.make_bags(4321, B, N, L)
    x = x32.reshape(B * N, L).to(dev).to(torch.bfloat16)
    y = syn.make_labels(99, B).to(dev)
    lay = BagLayout.uniform(B, N, dev)
    # parity of THIS batch, before any update: 2 of the 32 bags against the oracle on the same bf16-rounded inputs
    prob, z = tr.forward(x, lay, None)
    pr = dict(p)
    for k in ("aggregator.attention_V.0.weight", "aggregator.attention_U.0.weight"):
        pr[k] = p[k].to(torch.bfloat16).float()
    dl, top1 = 0.0, True
    for b in (0, B - 1):
        o = orc.image_only_forward(x32[b].to(torch.bfloat16).float(), pr)
        dl = max(dl, float((z[b].cpu() - o["logits"][0]).abs().max()))
        top1 = top1 and bool(torch.equal(prob[b].cpu().argmax(-1), o["prob"][0].argmax(-1)))
    # gradient parity of the bf16-MFMA weight gradient (bf16 saved gates): bags 0 and B-1 as a 2-bag batch against the
    # oracle's gradients on the same rounded inputs (the bar asserted in tests/test_gpu_bf16.py is 1.2e-2 for the gate
    # parameters - 2^-9 relative rounding per bf16 factor -, 5e-4 for the fp32 head)
    xb = torch.cat([x32[0], x32[B - 1]], 0)
    y2 = torch.stack([y[0], y[B - 1]], 0)
    tr.forward(xb.to(dev).to(torch.bfloat16), BagLayout.uniform(2, N, dev), y2)
    tr.backward()
    _, _, _, og = orc.batch_loss_and_grads([x32[0].to(torch.bfloat16).float(), x32[B - 1].to(torch.bfloat16).float()], y2.cpu(), pr)
    rel = lambda a_, b_: float((a_.double() - b_.double()).norm() / (b_.double().norm() + 1e-30))      # noqa: E731
    gerr = {k: rel(tr.fp.g(k).cpu(), og[k]) for k in og if float(og[k].norm()) > 1e-7}
    gate_err = max(v for k, v in gerr.items() if k.startswith("aggregator.attention"))
    head_err = max(v for k, v in gerr.items() if k.startswith("fc."))
    for _ in range(30):
        tr.forward(x, lay, y)
    for _ in range(warmup):
        tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(x, lay, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    R = B * N
    kb, _ = tr.time_step_groups(x, lay, y, 30)
    kb_alone = tr.time_pieces(x, lay, y, 20)
    flops = 4.0 * R * L * D_GATE
    # in-step groups: the weight gradient's launch pair (+ head parameter gradients) is one entry point on this path
    # round 4: the deep forward carries the pool partial pass in its epilogue (group "gate_fwd_with_pool_fused")
    fwd_key = "gate_fwd_with_pool_fused" if "gate_fwd_with_pool_fused" in kb else "gate_fwd"
    dom = max((fwd_key, "gate_bwd_dw_reduce_head_adam"), key=lambda k: kb[k])
    hbm = R * L * 2 / (kb[fwd_key] * 1e-3) / 1e9
    # the same step in model.train() mode (in-kernel dropout through the keep-bit tensors): THIS is what the object is
    # quoted on, like the headline; the eval-mode step and its launch groups - the ones the roofline objects
    # below are computed from - stand beside it
    del tr
    tr_t = ImageOnlyTrainer(p, dev, train_mode=True)
    for _ in range(warmup + 3):
        tr_t.train_step(x, lay, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // 2):
        tr_t.train_step(x, lay, y)
    torch.cuda.synchronize()
    ms_train = (time.perf_counter() - t0) / (steps // 2) * 1e3
    kb_train, _ = tr_t.time_step_groups(x, lay, y, 20)
    del tr_t
    return {"workload": f"{B} bags x {N} x {L}, bf16 storage / fp32 accumulate, image-only fwd+BCE+bwd+Adam (BASELINE config 5)",
            "mode": "train (dropout 0.5 on patches + 0.25 before the head, keep-bit tensors)", "ms_per_step": round(ms_train, 4),
            "bags_per_s": round(B / (ms_train * 1e-3), 1), "dtype": "bf16",
            "eval_mode_ms_per_step": round(ms, 4), "eval_mode_bags_per_s": round(B / (ms * 1e-3), 1),
            "kernels_ms_train_mode": {k: round(v, 4) for k, v in kb_train.items()},
            "roofline_note": "roofline / roofline_step / kernels_ms below: the eval-mode step (eval_mode_ms_per_step)",
            "step_algorithmic_bytes": 2 * R * L * 2, "step_hbm_frac": round(2 * R * L * 2 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "roofline": {"bound": "mfma", "kernel": ("k_gate_fwd_bf16_deep" + (" (+ pool pass in its epilogue)" if fwd_key != "gate_fwd" else ""))
                         if dom == fwd_key else "k_gate_bwd_dw_bf16 (+ its fold)",
                         "achieved": round(flops / (kb[dom] * 1e-3) / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(flops / (kb[dom] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                         "flops_per_launch": flops, "ms_per_launch": round(kb[dom], 4), "timing": "HIP events inside the running step",
                         "traffic": _pmc_traffic("cfg5_" + ("gate_fwd" if dom == fwd_key else "gate_bwd_dw")),
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this workload, bytes per launch)"},
            "roofline_step": {"bound": "mfma", "achieved": round(2 * flops / (ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(2 * flops / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                              "flops_per_step": 2 * flops},
            "grad_rel_err": {"gate_params_max": gate_err, "head_params_max": head_err, "asserted_in_tests": "1.2e-2 / 5e-4",
                             "batch": "bags 0 and 31 as a 2-bag batch vs the oracle on the same bf16-rounded inputs"},
            "roofline_hbm_gate_fwd": {"bound": "hbm", "achieved": round(hbm, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(hbm / PEAK_HBM_GBS, 4), "bytes_per_launch": R * L * 2},
            "kernels_ms": {k: round(v, 4) for k, v in kb.items()},
            "kernels_ms_standalone": {k: round(v, 4) for k, v in kb_alone.items()},
            "parity": {"bags_checked": [0, B - 1], "max_abs_dlogit": dl, "top1_equal": top1,
                       "oracle": "fp32 oracle on the same bf16-rounded x and gate weights", "tolerance": 1e-3}}


def config3_fusion(dev, steps=30, warmup=4):
    """BASELINE config 3: 32 bags x 1024 patches x 768 dims + one 77-token note per bag through aggregator(args):
    fc_pathology -> CLIP ViT-B/32 text tower (frozen) -> fc_CI2Pth -> TwoWayTransformer -> multi-modal bag -> ABMIL ->
    head -> BCE -> backward -> FlatAdam; trainable part replayed from a hipGraph (frozen text tower outside)."""
    import torch
    from types import SimpleNamespace
    from mil_amd import synthetic as syn
    from mil_amd.model.utils import get_model
    from mil_amd.optim import FlatAdam
    from mil_amd.ops import backward as ops_backward
    from oracle import mil_oracle as orc
    B, N = 32, 1024
    args = SimpleNamespace(modality=["pathology"], model_pathology="ABMIL", model_CI="CLIP", aggregator="ABMIL",
                           num_classes=2, learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9, alignment_base="CI",
                           model_CT="resnetMC3_18", clip_layers=12, cache_text=0)
    torch.manual_seed(1234)
    model = get_model(args).to(dev).eval()          # eval: the parity mode (dropout off); gradients still flow
    x = syn.make_bags(1, B, N, 768).to(dev)
    ids = syn.make_token_ids(2, B, 1).to(dev)
    y = syn.make_labels(3, B).to(dev)
    # parity of THIS batch on 2 of the 32 bags (before any update)
    with torch.no_grad():
        prob, _ = model([x], ids)
        z = model.last_logits.detach().cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    dl, top1 = 0.0, True
    for b in (0, B - 1):
        with torch.no_grad():
            o = orc.fused_forward(x[b].cpu(), ids[b].cpu(), sd)
        dl = max(dl, float((z[b] - o["logits"][0]).abs().max()))
        top1 = top1 and bool(torch.equal(prob[b].cpu().argmax(-1), o["prob"][0].argmax(-1)))
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-5, weight_decay=1e-7, counted=True)
    with torch.no_grad():
        tfeat = model.clinic_extractor(ids)         # frozen tower: cached per note in training (dim1/CLIP.py cache_text)

    def gstep():
        model([x], ids, text_features=tfeat, labels=y)      # criterion(prob, y) of train_ddp.py:323-324 inside the module:
        loss = model.last_loss                              # pool + head + BCE run as one fused node
        ops_backward(loss)
        opt.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            gstep()
    torch.cuda.current_stream().wait_stream(side)
    opt.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gstep()
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the same step with the text tower INSIDE the loop, as model/dim1/CLIP.py:71-75 runs it (encode_text under no_grad every
    # step, no per-note cache): frozen tower eagerly (its launch geometry follows the notes' lengths), then the replay
    def cold():
        with torch.no_grad():
            tfeat.copy_(model.clinic_extractor(ids))
        g.replay()
    for _ in range(2):
        cold()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ncold = max(4, steps // 3)
    for _ in range(ncold):
        cold()
    torch.cuda.synchronize()
    ms_cold = (time.perf_counter() - t0) / ncold * 1e3
    R = B * N
    # algorithmic flops of the trainable path per step: fc_pathology fwd + dW (2 x 2*R*768*512),
    # ABMIL gate fwd + dW + dx (3 x 4*R*512*192); the absorbed one-token attention sites are HBM-bound streams
    flops = 2 * 2.0 * R * 768 * 512 + 3 * 4.0 * R * 512 * D_GATE
    # algorithmic HBM bytes of the step: every [R, E] stream once per pass that must produce or consume it (SURVEY 8d's
    # per-pass counting; E = 512 floats = 2 KiB per row, the raw features 768)
    row = 512 * 4
    n_par = sum(p_.numel() for p_ in model.parameters() if p_.requires_grad)
    alg_bytes = (R * (768 + 512) * 4                              # fc_pathology forward: x in, xi out
                 + R * row                                        # first block: absorbed pool over the keys
                 + 2 * (2 * R * row)                              # LayerNorm(keys + row) fused with the NEXT site's pool (round 4): x in, keys out
                 + R * row + R * 384 * 4 + R * row                # gate forward (x0 in, gates out) + pool pass
                 + (R * row + R * 384 * 4) + (R * 384 * 4 + R * row)      # gate dW (x0, gates in); gate dx (gates in, dx out)
                 + (R * row + 3 * R * row)                        # first block's pool backward: dots (keys), apply (keys, dkeys in, dkeys out)
                 + 2 * (3 * R * row)                              # fused pair backward x 2, ONE pass: x, dy in, dx out (keys recomputed)
                 + R * (512 + 512 + 768) * 4                      # fc_pathology parameter backward: dy, y (tanh'), x
                 + n_par * 28)                                    # Adam: p, m, v in and out + g in
    return {"workload": f"{B} bags x {N} x 768 + one 77-token note per bag, aggregator(args) fwd+BCE+bwd+Adam "
                        "(BASELINE config 3; text embeddings of the frozen ViT-B/32 tower cached per note)",
            "ms_per_step": round(ms, 4), "bags_per_s": round(B / (ms * 1e-3), 1), "dtype": "f32", "launch": "hipGraph",
            "cold_ms_per_step": round(ms_cold, 4), "cold_bags_per_s": round(B / (ms_cold * 1e-3), 1),
            "cold_note": "encode_text of the 32 notes inside every step (no text cache; reference model/dim1/CLIP.py:71-75), "
                         "frozen tower eager + trainable part replayed",
            "roofline": {"bound": "mfma", "kernel": "step (fc_pathology + gate GEMMs)", "achieved": round(flops / (ms * 1e-3) / 1e12, 2),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                         "flops_per_step": flops, "traffic": _pmc_traffic("cfg3_step"),
                         "traffic_source": "profiles/pmc_traffic.json: HBM bytes of the WHOLE step (2 x FETCH_SIZE + WRITE_SIZE summed "
                                           "over every kernel of the replayed step, separate rocprofv3 --pmc passes)"},
            "step_hbm_ms_at_peak": (round(_pmc_traffic("cfg3_step") / (PEAK_HBM_GBS * 1e9) * 1e3, 4)
                                    if _pmc_traffic("cfg3_step") else None),
            "step_algorithmic_bytes": int(alg_bytes),
            "traffic_over_algorithmic": (round(_pmc_traffic("cfg3_step") / alg_bytes, 3) if _pmc_traffic("cfg3_step") else None),
            "step_algorithmic_note": "every [32768, 512] / [32768, 768] stream once per pass that produces or consumes it + Adam's "
                                     "28 B per trainable parameter",
            "parity": {"bags_checked": [0, B - 1], "max_abs_dlogit": dl, "top1_equal": top1,
                       "oracle": "fp32 oracle fused_forward (model/aggregator.py:134-209 wiring)", "tolerance": 1e-3}}


# ----------------------------------------------------------------------------------------------- multi-GPU floors at N = 1
def rccl_floor(args):
    """What the N > 1 step costs before any link is involved: this same workload at world size 1 with the step's RCCL
    all-reduce forced in (MIL_FORCE_COLLECTIVES=1) - once with the collective issued eagerly between the fold and the Adam
    launch (today's N > 1 path) and once with [fold -> all_reduce -> Adam] captured in the step's hipGraph
    (MIL_GRAPH_COLLECTIVE=1).  Each is a child process (a process group cannot be added to a rank that has already run); a
    child that fails reports its error instead of a number."""
    out = {}
    base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--regions", "5", "--prime", str(args.prime), "--batches", str(args.batches), "--train-mode", str(args.train_mode),
            "--no-configs", "--no-cpu-baseline", "--no-breakdown"]
    for name, extra in (("eager_collective", {}), ("graph_collective", {"MIL_GRAPH_COLLECTIVE": "1"})):
        env = dict(os.environ, MIL_FORCE_COLLECTIVES="1", MASTER_PORT=str(_free_port()), **extra)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=300)
            ln = json.loads(r.stdout.strip().splitlines()[-1])
            out[name] = {"ms_per_step": ln["ms_per_step"], "ms_per_step_runs": ln["ms_per_step_runs"],
                         "allreduce_us": ln["rccl"]["allreduce_us"], "allreduce_bytes": ln["rccl"]["allreduce_bytes"],
                         "launch": ln["config"]["launch"]}
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    out["note"] = ("world size 1, RCCL all-reduce of the step's 0.79 MB buffer forced into every step: the fixed cost of the N > 1 "
                   "path (split fold / Adam launches + one collective call) beside the headline step, which folds Adam into the "
                   "fold launch; the first SCALE run adds the xGMI hops on top of these")
    return out


# ----------------------------------------------------------------------------------------------- the authors' regime (N=1)
def ragged_regime(dev, kind, n_bags=64, steps=None, host_steps=24):
    """One ragged bag per step, N ~ U[2000, 15592] changing every step (reference run_train.sh:81: --batch_size = number of
    GPUs; dataset.py:366-393: bags of up to 15 592 patches, a fresh 10 - 20 % patch drop every epoch), model.train().
    kind: "image" (image-only fused step, F = 512), "fusion" (aggregator(args): pathology + one note, ViT-B/32 text embedding
    cached per note), "ct_pth" (the authors' own run: CT + pathology, loss_point CT-Pth-Last + textCosSim).
    Three ways of feeding the SAME replayed per-bucket graphs, each timed end to end over whole epochs of the cohort:
      resident   cohort.DeviceCohort: bags in HBM, the epoch's drop drawn on the device, one gather launch per step
                 (`ms_per_step`, `feed`: what train_ddp.py --hip_graph 1 runs)
      replay     the step alone: the bag already on the device, one D2D copy into the bucket (round 3's figure)
      host       round 3's train_ddp loop: host bag -> zero-padded [1, n, F] pageable tensor -> .to(dev) -> D2D into the bucket
    Parity: two bags of the cohort, eval mode, fed through the cohort, against the oracle on the rows oracle/cohort.py says
    the device drew."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from mil_amd import synthetic as syn
    from mil_amd.cohort import DeviceCohort, keep_count
    from oracle import cohort as oc
    from oracle import mil_oracle as orc
    fusion = kind != "image"
    with_ct = kind == "ct_pth"
    F = 768 if fusion else 512
    steps = steps or (3 * n_bags if not fusion else 2 * n_bags)
    rng = np.random.default_rng(0)
    ns = [int(v) for v in rng.integers(2000, 15593, size=n_bags)]
    keeps = [0.9 if i % 2 else 0.8 for i in range(n_bags)]            # biopsies / resections (dataset.py:374-381)
    off = np.concatenate([[0], np.cumsum(ns)])
    big = torch.randn((int(off[-1]), F), device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    labels = syn.make_labels(3, n_bags)
    ids = syn.make_token_ids(2, n_bags, 1)
    co = DeviceCohort(lambda j: big[off[j]:off[j + 1]], labels, dev, ids=ids, keep=keeps, seed=1234, lengths=ns)
    CT_SHAPE = (512, 160, 2, 2)
    ct = syn.make_ct_map(5, 1, CT_SHAPE[1], CT_SHAPE[2]).to(dev) if with_ct else None

    if not fusion:
        from mil_amd.trainer import ImageOnlyTrainer, RaggedImageOnlyStepper
        p = syn.image_only_params(1, L=F)

        def make(train=True):
            tr = ImageOnlyTrainer(p, dev, train_mode=train, counted=True, lr=1e-5 if train else 0.0)    # parity pass: weights stay put
            return tr, RaggedImageOnlyStepper(tr, B=1)
        tr, st = make()

        def feed_step(st_, j):
            k = co.lengths([j])[0]
            slot = st_.slot(k)
            ks = co.feed([j], slot.x, slot.layout.bag_len_dev, slot.y)
            return st_.step(slot, ks, on_device=True)

        def put_step(st_, xrows, yrow, k):
            slot = st_.slot(k)
            slot.x[:k].copy_(xrows, non_blocking=True)
            slot.y.copy_(yrow, non_blocking=True)
            return st_.step(slot, [k])
        graphs = lambda st_: sum(s_.graph is not None for s_ in st_.slots.values())      # noqa: E731
    else:
        from mil_amd.fusion_step import RaggedFusionStepper
        from mil_amd.model.utils import get_model
        from mil_amd.optim import FlatAdam
        args = SimpleNamespace(modality=["CT", "pathology"] if with_ct else ["pathology"], model_pathology="ABMIL", model_CI="CLIP",
                               aggregator="ABMIL", num_classes=2, learnablePrompt=0, n_ctx=8, clinical_features=["f"] * 9,
                               alignment_base="CI", model_CT="resnetMC3_18", clip_layers=12, cache_text=1)
        torch.manual_seed(1234)
        model = get_model(args).to(dev)
        with torch.no_grad():
            co.set_text(torch.cat([model.clinic_extractor(co.ids[i:i + 16]) for i in range(0, n_bags, 16)], 0))

        def make(train=True):
            model.train(train)
            opt = FlatAdam([q for q in model.parameters() if q.requires_grad], lr=1e-5 if train else 0.0, weight_decay=1e-7,
                           counted=True)
            return opt, RaggedFusionStepper(model, opt, B=1, ct_shape=CT_SHAPE if with_ct else None,
                                            loss_mult=3.0 if with_ct else 1.0, cossim=with_ct)
        tr, st = make()

        def feed_step(st_, j):
            k = co.lengths([j])[0]
            slot = st_.slot(k)
            ks = co.feed([j], slot.x, slot.bucket.len_dev, slot.y, text_dst=slot.text)
            if with_ct:
                slot.ct.copy_(ct, non_blocking=True)
            return st_.step(slot, ks, on_device=True)

        def put_step(st_, xrows, yrow, k, j=0):
            slot = st_.slot(k)
            slot.x[:k].copy_(xrows, non_blocking=True)
            slot.y.copy_(yrow, non_blocking=True)
            slot.text.copy_(co.text[j:j + 1], non_blocking=True)
            if with_ct:
                slot.ct.copy_(ct, non_blocking=True)
            return st_.step(slot, [k])
        graphs = lambda st_: len(st_.gs._graphs)      # noqa: E731

    def epoch_order(e):
        return [int(v) for v in np.random.default_rng(100 + e).permutation(n_bags)]

    # ---- resident cohort, end to end: draw per epoch + feed per step + replay
    done, e = 0, 0
    while done < n_bags:                              # one warm epoch: every bucket captured
        co.draw_epoch(e)
        for j in epoch_order(e):
            feed_step(st, j)
            done += 1
        e += 1
    torch.cuda.synchronize()
    t0, done = time.perf_counter(), 0
    while done < steps:
        co.draw_epoch(e)
        for j in epoch_order(e):
            if done >= steps:
                break
            feed_step(st, j)
            done += 1
        e += 1
    torch.cuda.synchronize()
    ms_res = (time.perf_counter() - t0) / steps * 1e3
    n_graphs = graphs(st)
    # ---- the step alone (device-resident bag, one D2D copy): comparable with round 3's 0.149 / 1.07 / 1.8 ms
    seq = [epoch_order(e + 1)[i % n_bags] for i in range(steps)]
    ylab = co.labels
    for j in seq[:8]:
        put_step(st, big[off[j]:off[j] + co.k_train[j]], ylab[j:j + 1], co.k_train[j], *([j] if fusion else []))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in seq:
        put_step(st, big[off[j]:off[j] + co.k_train[j]], ylab[j:j + 1], co.k_train[j], *([j] if fusion else []))
    torch.cuda.synchronize()
    ms_replay = (time.perf_counter() - t0) / steps * 1e3
    # ---- round 3's loop: the bag comes from pageable host memory every step
    host_bags = {j: big[off[j]:off[j] + co.k_train[j]].cpu() for j in seq[:host_steps]}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in seq[:host_steps]:
        xb = torch.zeros((1, co.k_train[j], F))        # collate_bags' zero-pad to [B, maxN, F] (dataset.py:386-391)
        xb[0] = host_bags[j]
        xd = xb.to(dev, non_blocking=True)              # pageable: the copy is synchronous
        put_step(st, xd[0], labels[j:j + 1].to(dev, non_blocking=True), co.k_train[j], *([j] if fusion else []))
    torch.cuda.synchronize()
    ms_host = (time.perf_counter() - t0) / host_steps * 1e3
    # ---- the fallback for a cohort that does not fit HBM (cohort.HostFeed): a background thread fills two pinned staging
    # buffers with the UN-dropped bag, the copy runs on its own stream, drop + placement on the device as above
    from mil_amd.cohort import HostFeed
    full_host = {j: big[off[j]:off[j + 1]].cpu().pin_memory() for j in set(seq[:host_steps + 1])}     # the cohort in PINNED host RAM
    hf = HostFeed(lambda j: full_host[j], ns, F, labels, dev, ids=ids, keep=keeps, seed=co.seed)

    def hf_step(j, nxt):
        k = co.k_train[j]
        slot = st.slot(k)
        ldev = slot.bucket.len_dev if fusion else slot.layout.bag_len_dev
        hf.next(slot.x, ldev, slot.y, epoch=e)
        if nxt is not None:
            hf.prefetch(nxt)                             # loads while this step runs
        if fusion:
            slot.text.copy_(co.text[j:j + 1], non_blocking=True)
            if with_ct:
                slot.ct.copy_(ct, non_blocking=True)
        return st.step(slot, [k], on_device=True)
    # three passes over the same bags, the median: the leg runs at the mercy of the host's thread scheduling (one pass of a
    # profile round read 4.7 ms where every other run reads 1.0)
    runs_pinned = []
    for _ in range(3):
        hf.prefetch(seq[0])
        hf_step(seq[0], seq[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, host_steps + 1):
            hf_step(seq[i], seq[i + 1] if i < host_steps else None)
        torch.cuda.synchronize()
        runs_pinned.append((time.perf_counter() - t0) / host_steps * 1e3)
    ms_pinned = sorted(runs_pinned)[1]
    del host_bags, full_host, hf
    # ---- parity: eval mode, two bags fed through the cohort vs the oracle on the rows the numpy restatement selects
    tr_e, st_e = make(train=False)
    if fusion:
        sd = {k_: v_.detach().cpu() for k_, v_ in model.state_dict().items()}
    co.draw_epoch(10 ** 6)
    dl, top1 = 0.0, True
    for j in (0, n_bags - 1):
        out = feed_step(st_e, j)
        rows = oc.patch_drop_select(ns[j], keep_count(ns[j], keeps[j]), j, co.seed, 10 ** 6)
        xb = big[off[j]:off[j + 1]].cpu()[rows]
        with torch.no_grad():
            if not fusion:
                o = orc.image_only_forward(xb, p)
                z, pr = tr_e.last["logits"].cpu(), out[1].cpu()
            elif with_ct:
                o = orc.fused_forward_ct_pth(ct[0].cpu(), xb, ids[j], sd)
                z, pr = out[2].cpu(), out[1].cpu()
            else:
                o = orc.fused_forward(xb, ids[j], sd)
                z, pr = out[2].cpu(), out[1].cpu()
        dl = max(dl, float((z - o["logits"]).abs().max()))
        top1 = top1 and bool(torch.equal(pr.argmax(-1), o["prob"].argmax(-1)))
    mean_k = float(np.mean(co.k_train))
    bytes_step = mean_k * F * 4
    work = {"image": f"image-only fused step (fwd + BCE + bwd + Adam), F = {F}",
            "fusion": "aggregator(args): fc_pathology + cached ViT-B/32 note embedding + TwoWay fusion + ABMIL + head, fwd + BCE + bwd + Adam",
            "ct_pth": "aggregator(args), modality [CT, pathology], CT feature map [512, 160, 2, 2] as input, loss_point CT-Pth-Last "
                      "+ textCosSim (run_train.sh:81), fwd + loss + bwd + Adam"}[kind]
    res = {"workload": f"1 ragged bag per step, {n_bags}-bag cohort N ~ U[2000, 15592] (mean {np.mean(ns):.0f}, kept {mean_k:.0f} "
                       f"after the per-epoch 10/20 % drop) x {F}, model.train(); {work}",
           "ms_per_step": round(ms_res, 4), "bags_per_s": round(1e3 / ms_res, 1), "steps": steps, "graphs": n_graphs,
           "feed": "HBM-resident cohort: per-epoch patch drop drawn on the device (mil_patch_drop_select), one gather launch per "
                   "step into the bucket (mil_cohort_feed)",
           "replay_only_ms_per_step": round(ms_replay, 4), "feed_overhead": round(ms_res / ms_replay - 1.0, 4),
           "from_host_pageable_ms_per_step": round(ms_host, 4),
           "from_host_note": f"{host_steps} steps of round 3's loop: zero-padded pageable host bag -> .to(dev) -> D2D into the bucket",
           "from_host_pinned_prefetch_ms_per_step": round(ms_pinned, 4),
           "from_host_pinned_prefetch_runs": [round(v, 4) for v in runs_pinned],
           "from_host_pinned_note": "cohort.HostFeed, the fallback when the cohort does not fit HBM: the cohort in pinned host memory, "
                                    "the un-dropped bag of the NEXT step copied H2D on a copy stream into a device double buffer "
                                    "while this step runs, drop and placement on the device (PCIe-bound: ~45 MB per bag)",
           "feed_bytes_per_step": int(2 * bytes_step), "feed_bytes_note": "gather: kept rows read once from the cohort + written once to the bucket",
           "parity": {"bags_checked": [0, n_bags - 1], "max_abs_dlogit": dl, "top1_equal": top1, "tolerance": 1e-3,
                      "oracle": "eval mode; oracle forward on the rows oracle/cohort.py selects for the same (seed, epoch, bag)"}}
    del co, big
    return res


# ----------------------------------------------------------------------------------------------- one rank
def run_rank(args):
    import torch
    import torch.distributed as dist
    import mil_amd  # noqa: F401
    from mil_amd import synthetic as syn
    from mil_amd.bags import BagLayout
    from mil_amd.trainer import ImageOnlyTrainer

    strict_fail = False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                         f"(torchrun --nproc-per-node {args.gpus}, or plain `python bench.py --gpus {args.gpus}`)")
    # MIL_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo - a one-GPU rehearsal of the N>1 code path
    # (RCCL refuses two ranks on one device); never used for reported numbers.
    rehearsal = os.environ.get("MIL_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    if not rehearsal and local_rank >= ndev:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {ndev} are visible")
    gpu = 0 if rehearsal else local_rank
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    force = os.environ.get("MIL_FORCE_COLLECTIVES") == "1"      # world size 1 through RCCL: checks init + collectives
    use_dist = world > 1 or force
    backend = "gloo" if rehearsal else "nccl"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        kw = {} if rehearsal else {"device_id": dev}
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=300), **kw)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: {dist.get_world_size()} ranks joined, {args.gpus} expected")

    B, N, L, C = args.bags_per_gpu, args.patches, args.dim, 2
    strong = args.global_bags > 0
    if strong:
        if args.global_bags % world:
            raise SystemExit(f"bench.py: --global-bags {args.global_bags} is not a multiple of the {world} ranks")
        B = args.global_bags // world
    params = syn.image_only_params(1234, L=L)
    # MIL_GRAPH_COLLECTIVE=1 (world size > 1 or forced collectives): the step - all-reduce included - replays from one hipGraph
    # per resident batch, [forward .. fold -> all_reduce -> Adam] as one captured segment (trainer.capture)
    graph_coll = use_dist and os.environ.get("MIL_GRAPH_COLLECTIVE") == "1" and args.accum == 1 and not rehearsal
    tr = ImageOnlyTrainer(params, dev, world_size=world, train_mode=bool(args.train_mode), accum=args.accum,
                          counted=bool(graph_coll) or bool(args.graph))      # a captured step keeps its step number / mask position on the device
    nb = max(1, args.batches) if not args.graph else 1       # a captured step replays its static buffers
    xs = [syn.make_bags(4321 + rank + 1000 * i, B, N, L).reshape(B * N, L).to(dev) for i in range(nb)]   # resident in HBM before timing
    if args.dtype == "bf16":
        xs = [t.to(torch.bfloat16) for t in xs]
    ys = [syn.make_labels(99 + rank + 1000 * i, B, C).to(dev) for i in range(nb)]
    x, y = xs[0], ys[0]
    lay = BagLayout.uniform(B, N, dev)

    done_ev = torch.cuda.Event()

    def barrier():
        # synchronize on both sides of the rank barrier.  The wait itself polls an event recorded behind the last launch (the
        # blocking wait of hipDeviceSynchronize wakes tens of microseconds after the GPU has finished - 1 % of a 20-step region);
        # torch.cuda.synchronize() then returns at once and keeps the contract's semantics
        done_ev.record()
        while not done_ev.query():
            pass
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if args.dump:
        tr.forward(x, lay, y)
        tr.backward()
        tr.reduce_only()
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"grad": tr.fp.grad.detach().cpu().clone(), "loss": float(tr.loss_sum.item())}, args.dump)
        tr.reset_dropout_stream()

    launch_kind = "eager" if not args.graph else "hipGraph(fwd+bwd)+eager(allreduce,adam)"
    graphs = None
    if graph_coll and not args.graph:
        try:
            graphs = []
            for i in range(nb):
                tr.capture(xs[i], lay, ys[i], collective_in_graph=True)
                graphs.append(tr._graph)
            if not all(g["collective"] for g in graphs):
                graphs = None
        except Exception as e:      # noqa: BLE001  (capability probe: RCCL / torch may refuse the capture)
            print(f"bench.py: collective-in-graph capture refused ({type(e).__name__}: {e}); eager collective", file=sys.stderr)
            graphs = None
    if graphs is not None:
        turn = [0]
        launch_kind = "hipGraph(fwd+bwd+allreduce+adam), one per resident batch"

        def step():
            i = turn[0] % nb
            turn[0] += 1
            tr._graph = graphs[i]
            return tr.replay_step()
    elif not args.graph:
        turn = [0]

        def step():
            i = turn[0] % nb
            turn[0] += 1
            return tr.train_step(xs[i], lay, ys[i])
    else:
        tr.capture(x, lay, y)                        # forward+backward as one hipGraph on static buffers
        step = tr.replay_step
    # Clock priming (every rank, before the W warm-up steps): the chip needs ~10-30 ms of sustained load to reach its
    # steady clock, and a short run (50 steps = 13 ms) otherwise reads 8 % slower than a long one of the very same loop.
    # Forward passes only: no optimizer update, so the parameter trajectory of "W warm-up + K timed steps" is untouched.
    for i in range(args.prime):
        tr.forward(xs[i % nb], lay, ys[i % nb])
    for _ in range(args.warmup):
        step()
    # test hook (tests/test_gpu_bench_launcher.py): this rank dies between warm-up and the timed region, the others are
    # left waiting in the barrier - the launcher must notice, stop them and return non-zero
    if os.environ.get("MIL_BENCH_FAIL_RANK") == str(rank):
        os._exit(3)
    # EXACTLY --steps steps between a barrier + synchronize on both sides - and that region --regions times back to back: at
    # the driver's flags one region is 20 steps = 5 ms, where a single 0.3 ms host hiccup moves the figure by 6 % (round 3's
    # driver run read 8 % below every other run of the same command); the median region is what the line reports
    regions = []
    for _ in range(max(1, args.regions)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        regions.append(time.perf_counter() - t0)
    if use_dist:
        t = torch.tensor(regions, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)            # per region: the slowest rank
        regions = [float(v) for v in t.tolist()]
    elapsed = sorted(regions)[len(regions) // 2]
    loss = float(tr.loss_sum.item())

    rccl = None
    if use_dist:
        # the step's one collective, alone: the flat gradient + loss buffer (0.79 MB), HIP events on the compute stream
        # (torch's all_reduce makes the current stream wait for the collective, so the events bracket it)
        buf = torch.zeros_like(tr.fp.grad_ext)
        ar_ms = timed(lambda: dist.all_reduce(buf), 50, warm=10)
        t = torch.tensor([ar_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ver = None
        if backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:       # noqa: BLE001
                ver = None
        ar_us = float(t.item()) * 1e3
        rccl = {"backend": backend, "world_size": dist.get_world_size(), "allreduce_bytes": buf.numel() * 4,
                "allreduce_us": round(ar_us, 2), "version": ver,
                "collectives_per_step": round(1.0 / args.accum, 4),
                # nothing of the step is left to overlap the collective with (the buffer is complete when the fold launch
                # ends), so all of it is exposed: its share of the step as measured
                "exposed_fraction_of_step": round(ar_us / args.accum / (elapsed / args.steps * 1e6), 4)}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        mode = "train (dropout 0.5 on patches + 0.25 before the head, in-kernel masks)" if args.train_mode else "eval (no dropout)"
        line = {
            "metric": "bags/sec fwd+bwd, N=1024 patches D=512", "value": round(value, 1), "unit": "bags/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "ms_per_step_runs": [round(r / args.steps * 1e3, 4) for r in regions],
            "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4), "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
            "timing": f"median of {len(regions)} timed regions of exactly {args.steps} steps each (barrier + synchronize on both "
                      "sides, max over ranks per region)",
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{B} bags/GPU x {N} patches x {L} dims, image-only gated-attention MIL "
                                   f"fwd+BCE+bwd+allreduce+Adam (BASELINE config 2; x{world} GPUs = {world * B} bags"
                                   + (", strong scaling: the global batch is fixed - SURVEY 8d's config-4 reading" if strong else "") + ")",
                       "bags_per_gpu": B, "patches": N, "dim": L, "global_bags": world * B,
                       "parallelism": f"dp{world}", "loss": round(loss, 6), "mode": mode, "accum": args.accum,
                       "batches_cycled": nb,
                       "launch": launch_kind},
        }
        if rccl is not None:
            line["rccl"] = rccl
        R = B * N
        gate_flops = 4.0 * R * L * D_GATE
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        line["roofline_step"] = {"bound": "mfma", "achieved": round(2 * gate_flops / (ms_step * 1e-3) / 1e12, 2), "peak": peak,
                                 "unit": "TFLOP/s", "frac": round(2 * gate_flops / (ms_step * 1e-3) / 1e12 / peak, 4),
                                 "flops_per_step": 2 * gate_flops,
                                 "note": "gate GEMMs forward + weight gradient (2 x 4 R L D) over the driver-timed ms_per_step"}
        if not args.no_breakdown:
            # each launch group of the step, timed INSIDE the running step (HIP events between the groups, on the stream the
            # kernels are launched on; mil_image_only_step_profile) - the figure rocprofv3's kernel trace of the step reports
            rot = list(zip(xs, ys)) if nb > 1 else None       # the same rotation as the timed loop
            kb, ev_step = tr.time_step_groups(x, lay, y, 50, rot=rot)
            kb_alone = kernel_breakdown(tr, x, lay, y)
            names = {"gate_fwd_with_pool_fused": "k_gate_fwd2<train, pool pass in the epilogue>", "gate_fwd": "k_gate_fwd2",
                     "gate_bwd_dw": "k_gate_bwd_dw2", "gate_bwd_dw_reduce_head": "k_gate_bwd_dw_bf16 (+ fold)", "gate_bwd_dw_reduce_head_adam": "k_gate_bwd_dw_bf16 (+ fold with Adam)"}
            if args.dtype == "bf16":
                names.update(gate_fwd="k_gate_fwd_bf16")
            flops = {k: gate_flops for k in names if k in kb}
            dom = max(flops, key=lambda k: kb[k])
            ach = flops[dom] / (kb[dom] * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": names[dom], "group": dom, "achieved": round(ach, 2), "peak": peak,
                                "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                "traffic": _pmc_traffic("cfg2_" + ("gate_fwd" if dom.startswith("gate_fwd") else "gate_bwd_dw"))
                                if (B, N, L, args.dtype) == (32, 1024, 512, "f32") else None,
                                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this workload, bytes per launch)",
                                "flops_per_launch": flops[dom], "ms_per_launch": round(kb[dom], 4),
                                "timing": "HIP events inside the running step, 50 steps right after the timed region, rotating "
                                          f"through the same {nb} batch(es) as the timed loop"}
            line["kernels_ms"] = {k: round(v, 4) for k, v in kb.items()}
            line["kernels_ms_sum"] = round(sum(kb.values()), 4)
            line["kernels_ms_event_step"] = round(ev_step, 4)
            # the step as the events see it (its launches + the event gaps) against the host-timed median: they describe the
            # same loop on the same batches, so more than 5 % apart means one of the two measurements is off
            diff = abs(ev_step - ms_step) / ms_step
            line["consistency"] = {"ms_per_step": round(ms_step, 4), "kernels_ms_event_step": round(ev_step, 4),
                                   "kernels_ms_sum": round(sum(kb.values()), 4), "rel_diff": round(diff, 4), "ok": diff <= 0.05}
            if diff > 0.05:
                print(f"bench.py: WARNING - the host-timed step ({ms_step:.4f} ms) and the event-timed step ({ev_step:.4f} ms) "
                      f"differ by {diff:.1%} (> 5 %): the roofline objects do not describe the headline", file=sys.stderr)
            line["kernels_ms_standalone"] = {k: round(v, 4) for k, v in kb_alone.items()}
            if dom == "gate_fwd_with_pool_fused" and kb_alone.get("gate_fwd") and kb_alone.get("gate_fwd_with_pool_fused"):
                # the dominant launch is two phases: the gate GEMMs (MFMA-bound) and the pool partial pass of its epilogue (a
                # 64 MiB re-read: HBM / Infinity-Cache-bound).  Their split from the stand-alone entry points (one batch
                # repeated back to back, i.e. cache-resident x): what the matrix loop alone reaches against the same peak.
                g_, f_ = kb_alone["gate_fwd"], kb_alone["gate_fwd_with_pool_fused"]
                line["roofline"]["phases_standalone"] = {
                    "gate_gemm_ms": round(g_, 4), "gate_gemm_frac_of_peak": round(gate_flops / (g_ * 1e-3) / 1e12 / peak, 4),
                    "pool_epilogue_ms": round(max(f_ - g_, 0.0), 4),
                    "note": "stand-alone launches on one repeated batch; `frac` above is the fused launch inside the running step"}
            line["kernels_ms_note"] = ("kernels_ms: launch groups of the step as it runs (world size 1: keep bits drawn by the "
                                       "forward launch, pool partial pass in its epilogue, Adam inside the reduce launch); "
                                       "kernels_ms_standalone: the same entry points repeated back to back on their own")
            line["kernels_tflops"] = {k: round(flops[k] / (kb[k] * 1e-3) / 1e12, 2) for k in flops}
            if args.train_mode and args.dtype == "f32" and world == 1:
                # the same launches without dropout (model.eval() arithmetic): the keep-bit selects of train mode are work
                # the 4 R L D flop count does not contain, so the MFMA fraction of the bare products is reported beside it
                from mil_amd.trainer import ImageOnlyTrainer
                tr_e = ImageOnlyTrainer(params, dev, train_mode=False)
                kbe, _ = tr_e.time_step_groups(x, lay, y, 30, rot=rot)
                line["roofline_eval_mode"] = {
                    "bound": "mfma", "kernel": names[dom], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "achieved": round(flops[dom] / (kbe[dom] * 1e-3) / 1e12, 2),
                    "frac": round(flops[dom] / (kbe[dom] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "ms_per_launch": round(kbe[dom], 4), "kernels_ms": {k: round(v, 4) for k, v in kbe.items()}}
                del tr_e
            if world == 1 and args.dtype == "f32":
                line["roofline_pool"] = pool_roofline(dev)
        if world == 1 and not use_dist and not args.no_breakdown and not args.no_rccl_floor and args.dtype == "f32":
            line["rccl_floor"] = rccl_floor(args)
        if world == 1 and not args.no_configs and args.dtype == "f32":
            cfgs = {}
            jobs = [("cfg3", config3_fusion), ("cfg5", config5_bf16)]
            if not args.no_ragged:
                jobs += [("ragged_image", lambda d: ragged_regime(d, "image")), ("ragged_fusion", lambda d: ragged_regime(d, "fusion")),
                         ("ragged_ct_pth", lambda d: ragged_regime(d, "ct_pth"))]
            for name, fn in jobs:
                try:
                    cfgs[name] = fn(dev)
                except Exception as e:      # noqa: BLE001  (the headline line must still be printed)
                    cfgs[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            line["configs"] = cfgs
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(N, L)
        print(json.dumps(line), flush=True)
        if os.environ.get("MIL_BENCH_STRICT") == "1" and not line.get("consistency", {}).get("ok", True):
            strict_fail = True
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if strict_fail:
        raise SystemExit(3)


def main():
    args = parse_args()
    if args.only_ragged:
        import torch
        import mil_amd  # noqa: F401
        torch.cuda.set_device(0)
        print(json.dumps(ragged_regime(torch.device("cuda", 0), args.only_ragged)), flush=True)
        return
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args))
    run_rank(args)


if __name__ == "__main__":
    main()
