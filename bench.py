#!/usr/bin/env python3
"""bags/sec, forward+backward(+all-reduce+Adam), N=1024 patches, D=512 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic bags resident in HBM:
per GPU 32 bags x 1024 patches x 512 dims (BASELINE config 2 at N=1; 8 GPUs x 32 = the 256-bag
config 4), image-only branch: gate scores (fp32 MFMA) -> attention pool -> head -> BCE ->
backward -> one flat-gradient RCCL all-reduce -> Adam.  Weak scaling: per-GPU work is fixed.

Rank 0 prints ONE JSON line.  Extra objects on it:
  roofline      dominant kernel (the gate GEMMs, fp32 MFMA bound), algorithmic flops / HIP-event time
  roofline_pool the HBM-bound attention-pool kernel at N=4096, D=512 (north_star's 30 % target)
  cpu_baseline  the CPU oracle (torch fp32, one bag per forward as the reference runs) on this host
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import mil_amd  # noqa: E402,F401
from mil_amd import ops, synthetic as syn  # noqa: E402
from mil_amd.bags import BagLayout  # noqa: E402
from mil_amd.trainer import ImageOnlyTrainer  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# HBM bytes per launch from rocprofv3 PMC passes of this same command at the default workload
# (profiles/r01_bench_hbm_traffic_pmc.csv: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction applied)
PMC_TRAFFIC_BYTES = {"gate_fwd": (70.3 + 48.1) * 2 ** 20, "gate_bwd_dw": (130.7 + 15.8) * 2 ** 20}
PEAK_HBM_GBS = 8000.0            # HBM3E spec
D_GATE = 192


def timed(fn, iters, warm=2):
    """Average duration (ms) of fn() on the current stream, measured with HIP events."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_breakdown(tr, x, lay, y, iters=20):
    """HIP-event time of each launch group of one step (separate pass, outside the timed region)."""
    fp = tr.fp
    R, L = x.shape
    tr.forward(x, lay, y)
    c = dict(tr.last)
    w = fp.p("aggregator.attention_weights.weight").view(-1)
    out = {}
    out["gate_fwd"] = timed(lambda: tr._gate_fwd(x, True), iters)
    out["pool_partial"] = timed(lambda: ops.attn_pool_partial_h(x, c["scores"], lay, fp.p("fc.1.weight")), iters)
    partials, hrow = ops.attn_pool_partial_h(x, c["scores"], lay, fp.p("fc.1.weight"))
    scale = 1.0 / c["prob"].numel()
    # with hrow the fused tail also writes the score gradient ds (the former k_pool_ds_from_h launch)
    out["merge_head_loss_ds"] = timed(lambda: ops.pool_merge_head(partials, lay, L, fp.p("fc.1.weight"), fp.p("fc.1.bias"),
                                                                  y, scale, scores=c["scores"], hrow=hrow), iters)
    ds = c["ds"]
    g = {k: torch.empty_like(fp.p(k)) for k in fp.order}
    gargs = (g["aggregator.attention_V.0.weight"], g["aggregator.attention_V.0.bias"], g["aggregator.attention_U.0.weight"],
             g["aggregator.attention_U.0.bias"], g["aggregator.attention_weights.weight"].view(-1),
             g["aggregator.attention_weights.bias"])
    ws = ops.gate_bwd_params(x, c["gates"], ds, w, *gargs)
    lib = ops._lib.lib()
    out["gate_bwd_dw"] = timed(lambda: lib.mil_gate_bwd_partials(ops._p(x), ops._p(c["gates"]), ops._p(ds), ops._p(w), R, L,
                                                                 D_GATE, ops._p(ws), ws.numel(), ops._stream()), iters)
    # the split-K fold; the head's parameter gradients (dWf, dbf, loss sum) ride on the same launch as appended workgroups
    both = timed(lambda: ops.gate_bwd_params_head(x, c["gates"], ds, w, *gargs, c["dz"], c["M"], g["fc.1.weight"], g["fc.1.bias"],
                                                  c["loss_bag"], tr.loss_sum, workspace=ws), iters)
    out["gate_bwd_reduce_and_head_params"] = max(0.0, both - out["gate_bwd_dw"])
    out["adam"] = timed(lambda: ops.adam_step(fp.flat, fp.grad, fp.exp_avg, fp.exp_avg_sq, 1), iters)
    return out


def pool_roofline(dev, iters=20):
    """The HBM-bound attention-pool stage at the north_star point N=4096, D=512: one pass reads x once
    (N*L*4 B), the scores (4 N) and writes M (4 L).  Enough bags to exceed the 256 MiB Infinity Cache
    so the bytes really come from HBM."""
    N, L, B = 4096, 512, 64            # 512 MiB of x
    x = torch.randn(B * N, L, device=dev)
    scores = torch.randn(B * N, device=dev)
    lay = BagLayout.uniform(B, N, dev)
    ms = timed(lambda: ops.attn_pool_fwd(x, scores, lay), iters, warm=3)
    alg_bytes = B * (N * L * 4 + 4 * N + 4 * L)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_pool_partial+k_pool_merge", "workload": f"{B} bags x {N} x {L} fp32",
            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4),
            "traffic": None, "ms_per_launch": round(ms, 4), "passes_per_s": round(B / (ms * 1e-3), 1)}


def cpu_baseline(N, L, budget_s=12.0):
    """The reference's CPU arithmetic (oracle restatement) timed on this host: one bag per forward,
    fp32, eval, fwd + BCE + bwd."""
    from oracle import mil_oracle as orc
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    p = syn.image_only_params(1234, L=L)
    names = list(p.keys())
    y = syn.make_labels(1, 1)
    bags = [torch.randn(N, L) for _ in range(4)]

    def one(xb):
        leaves = {k: p[k].clone().requires_grad_(True) for k in names}
        o = orc.image_only_forward(xb, leaves)
        loss = orc.bce_loss(o["prob"], y)
        loss.backward()

    for xb in bags[:3]:
        one(xb)
    n, t0 = 0, time.perf_counter()
    while True:
        one(bags[n % 4])
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {"value": round(n / el, 2), "unit": "bags/s", "cores": threads, "kind": "port",
            "sample": f"{n} bags of {N}x{L} fp32, one bag per fwd+loss+bwd, torch-CPU oracle, {el:.1f} s",
            "cpu": _cpu_model()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--prime", type=int, default=300, help="untimed forward passes before the warm-up steps (clock ramp)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bags-per-gpu", type=int, default=32)
    ap.add_argument("--patches", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: BASELINE config 5 variant (x and gate weights stored bf16, fp32 accumulate); use with "
                         "--patches 4096 --dim 1024.  The headline metric is f32.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward as one hipGraph (default: eager; "
                    "the step is GPU-bound either way: 0.306 vs 0.299 ms measured)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # MIL_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo - a one-GPU rehearsal of the N>1 code path
    # (RCCL refuses two ranks on one device); never used for reported numbers.
    rehearsal = os.environ.get("MIL_BENCH_REHEARSAL") == "1"
    gpu = 0 if rehearsal else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    force = os.environ.get("MIL_FORCE_COLLECTIVES") == "1"      # world size 1 through RCCL: checks init + collectives
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if force and world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    B, N, L, C = args.bags_per_gpu, args.patches, args.dim, 2
    params = syn.image_only_params(1234, L=L)
    tr = ImageOnlyTrainer(params, dev, world_size=world)
    x = syn.make_bags(4321 + rank, B, N, L).reshape(B * N, L).to(dev)       # resident in HBM before timing
    if args.dtype == "bf16":
        x = x.to(torch.bfloat16)
    y = syn.make_labels(99 + rank, B, C).to(dev)
    lay = BagLayout.uniform(B, N, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force:
            dist.barrier()
        torch.cuda.synchronize()

    if not args.graph:
        step = lambda: tr.train_step(x, lay, y)     # noqa: E731
    else:
        tr.capture(x, lay, y)                        # forward+backward as one hipGraph on static buffers
        step = tr.replay_step
    # Clock priming (every rank, before the W warm-up steps): the chip needs ~10-30 ms of sustained load to reach its
    # steady clock, and a short run (50 steps = 13 ms) otherwise reads 8 % slower than a long one of the very same loop
    # (0.287 vs 0.263 ms/step at 50 / 1000 steps).  Forward passes only: no optimizer update, so the parameter trajectory
    # of "W warm-up steps + K timed steps" is untouched.
    for _ in range(args.prime):
        tr.forward(x, lay, y)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1 or force:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(tr.loss_sum.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        line = {
            "metric": "bags/sec fwd+bwd, N=1024 patches D=512", "value": round(value, 1), "unit": "bags/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{B} bags/GPU x {N} patches x {L} dims, image-only gated-attention MIL "
                                   f"fwd+BCE+bwd+allreduce+Adam (BASELINE config 2; x{world} GPUs = {world * B} bags)",
                       "bags_per_gpu": B, "patches": N, "dim": L, "global_bags": world * B,
                       "parallelism": f"dp{world}", "loss": round(loss, 6),
                       "launch": "eager" if not args.graph else "hipGraph(fwd+bwd)+eager(allreduce,adam)"},
        }
        if not args.no_breakdown and args.dtype == "bf16":
            sb = 2                                    # bytes per stored x element
            fwd_ms = timed(lambda: tr._gate_fwd(x, True), 20)
            sc, _ = tr._gate_fwd(x, True)
            pp_ms = timed(lambda: ops.attn_pool_partial_bf16(x, sc, lay), 20)
            R = B * N
            line["roofline"] = {"bound": "hbm", "kernel": "k_gate_fwd_bf16", "achieved": round(R * L * sb / (fwd_ms * 1e-3) / 1e9, 1),
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(R * L * sb / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                "traffic": None, "ms_per_launch": round(fwd_ms, 4),
                                "mfma_tflops": round(4.0 * R * L * D_GATE / (fwd_ms * 1e-3) / 1e12, 1)}
            tr.forward(x, lay, y)
            c = dict(tr.last)
            ds = ops.attn_pool_bwd_bf16(x, c["scores"], c["lse"], c["dM"], c["cdot"], lay)
            ds_ms = timed(lambda: ops.attn_pool_bwd_bf16(x, c["scores"], c["lse"], c["dM"], c["cdot"], lay), 20)
            fp = tr.fp
            g = {k: torch.empty_like(fp.p(k)) for k in fp.order}
            gargs = (g["aggregator.attention_V.0.weight"], g["aggregator.attention_V.0.bias"],
                     g["aggregator.attention_U.0.weight"], g["aggregator.attention_U.0.bias"],
                     g["aggregator.attention_weights.weight"].view(-1), g["aggregator.attention_weights.bias"])
            wv = fp.p("aggregator.attention_weights.weight").view(-1)
            ws = ops.gate_bwd_params_bf16(x, c["gates"], ds, wv, *gargs)
            dw_ms = timed(lambda: ops.gate_bwd_params_bf16(x, c["gates"], ds, wv, *gargs, False, ws), 20)
            line["kernels_ms"] = {"gate_fwd_bf16": round(fwd_ms, 4), "pool_partial_bf16": round(pp_ms, 4),
                                  "pool_bwd_ds_bf16": round(ds_ms, 4), "gate_bwd_dw_bf16(+reduce)": round(dw_ms, 4)}
            line["kernels_gbs"] = {"pool_partial_bf16": round(R * L * sb / (pp_ms * 1e-3) / 1e9, 1),
                                   "pool_bwd_ds_bf16": round(R * L * sb / (ds_ms * 1e-3) / 1e9, 1)}
            line["kernels_tflops"] = {"gate_fwd_bf16": round(4.0 * R * L * D_GATE / (fwd_ms * 1e-3) / 1e12, 1),
                                      "gate_bwd_dw_bf16": round(4.0 * R * L * D_GATE / (dw_ms * 1e-3) / 1e12, 1)}
        elif not args.no_breakdown:
            kb = kernel_breakdown(tr, x, lay, y)
            R = B * N
            flops = {"gate_fwd": 4.0 * R * L * D_GATE, "gate_bwd_dw": 4.0 * R * L * D_GATE}
            dom = max(("gate_fwd", "gate_bwd_dw"), key=lambda k: kb[k])
            ach = flops[dom] / (kb[dom] * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": "k_gate_fwd" if dom == "gate_fwd" else "k_gate_bwd_dw",
                                "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                                "traffic": PMC_TRAFFIC_BYTES[dom] if (B, N, L) == (32, 1024, 512) else None,
                                "flops_per_launch": flops[dom], "ms_per_launch": round(kb[dom], 4)}
            line["kernels_ms"] = {k: round(v, 4) for k, v in kb.items()}
            line["kernels_tflops"] = {k: round(flops[k] / (kb[k] * 1e-3) / 1e12, 2) for k in flops}
            pool_bytes = R * L * 4 + 4 * R + 4 * L * B
            line["kernels_gbs"] = {"pool_partial": round(pool_bytes / (kb["pool_partial"] * 1e-3) / 1e9, 1),
                                   }
            line["roofline_pool"] = pool_roofline(dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(N, L)
        print(json.dumps(line), flush=True)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
