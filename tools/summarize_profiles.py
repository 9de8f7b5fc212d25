"""Turn gpurun_out/prof_round/ (tools/profile_round.sh) into the small tracked summaries under profiles/ and the
per-kernel HBM traffic table bench.py reads (profiles/pmc_traffic.json)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
CFG2_STEPS, CFG2_WARMUP = 200, 10
WORK = {"cfg2": f"python3 bench.py --no-configs --no-cpu-baseline --no-breakdown --regions 3 --steps {CFG2_STEPS} --warmup {CFG2_WARMUP}   "
                "(32 bags x 1024 x 512 fp32, train-mode step, default --prime 300; kernels of the step only)",
        "pool": "python3 tools/prof_pool.py   (attention-pool stage alone, 64 bags x 4096 x 512 fp32 = 512 MiB of x)",
        "cfg5": "python3 tools/prof_stage.py --bf16   (32 bags x 4096 x 1024, bf16 storage, 12 steps)",
        "cfg3": "python3 tools/bench_fusion.py --graph --steps 20 --warmup 3   (32 bags x 1024 x 768 + CLIP ViT-B/32 text, hipGraph replays)"}


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def stats(w, top=40):
    # gpurun merges a run's files into gpurun_out/ beside those of earlier runs: take the newest
    f = sorted(glob.glob(os.path.join(SRC, "stats_" + w, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))
    with open(os.path.join(DST, f"{TAG}_{w}_kernel_stats.csv"), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats -- {WORK[w]}   (MI355X, 1 GPU)\n")
        o.write("# one workload per file: every average below belongs to that shape.  Durations under the profiler read a few % "
                "above bench.py's HIP-event times.\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:top]:
            o.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.1f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")


def pmc(sub):
    f = sorted(glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def timed_region_stats():
    """cfg2: per-kernel averages over the dispatches of bench.py's TIMED region only (the 300 priming passes and the warm-up
    steps run while the clock ramps and are not what ms_per_step measures).  The region is found in the kernel trace itself:
    the weight-gradient kernel runs once per step, so its dispatches [warmup, warmup + steps) bracket the timed steps.
    Returns {kernel: (calls_per_step, avg_ns)} and the profiled run's own ms_per_step."""
    f = sorted(glob.glob(os.path.join(SRC, "stats_cfg2", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not f:
        return None, None
    rows = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f[0]))]
    rows.sort(key=lambda r: r[1])
    dw = [r for r in rows if r[0].startswith("k_gate_bwd_dw")]
    if len(dw) < CFG2_WARMUP + CFG2_STEPS:
        return None, None
    fwd = [r for r in rows if r[0].startswith("k_gate_fwd")]
    first_dw = dw[CFG2_WARMUP][1]
    t0 = max(r[1] for r in fwd if r[1] < first_dw)            # the forward launch that opens the first timed step
    t1 = dw[CFG2_WARMUP + CFG2_STEPS - 1][2]
    agg = collections.defaultdict(list)
    for n, a, b in rows:
        if a >= t0 and b <= t1 + 50000:                        # + the reduce launch that closes the last step
            agg[n].append(b - a)
    out = {n: (len(v) / CFG2_STEPS, sum(v) / len(v)) for n, v in agg.items()}
    ms = None
    try:
        txt = [l for l in open(os.path.join(SRC, "stats_cfg2.log")).read().splitlines() if l.startswith("{")][-1]
        ms = json.loads(txt)["ms_per_step"]
    except Exception:       # noqa: BLE001
        pass
    return out, ms


os.makedirs(DST, exist_ok=True)
for w in WORK:
    stats(w, top=60 if w == "cfg3" else 30)
region, prof_ms = timed_region_stats()
CHECK_FAILED = False
if region:
    try:
        txt = [l for l in open(os.path.join(SRC, "bench_line.json")).read().splitlines() if l.startswith("{")][-1]
        bl = json.loads(txt)
    except Exception:       # noqa: BLE001
        bl = {}
    ev = bl.get("kernels_ms", {})
    pair = {"k_gate_fwd": "gate_fwd_with_pool_fused", "k_gate_bwd_dw": "gate_bwd_dw", "k_pool_merge_head": "merge_head_loss_ds", "k_pool_tail_h": "merge_head_loss_ds",
            "k_gate_bwd_reduce": "gate_bwd_reduce_head_adam"}
    with open(os.path.join(DST, f"{TAG}_cfg2_timed_region.csv"), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace -- {WORK['cfg2']}\n")
        o.write("# averages over the dispatches INSIDE the timed region (the 200 steps the line's ms_per_step is measured on), from the\n"
                "# kernel trace; hip_event_us = the un-profiled bench.py run's in-step HIP-event time of the same launch (kernels_ms).\n")
        o.write("kernel,calls_per_step,rocprof_avg_us,hip_event_us,rocprof_over_event\n")
        tot = 0.0
        for n, (cps, avg) in sorted(region.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
            tot += cps * avg
            e = next((ev.get(v) for k, v in pair.items() if n.startswith(k)), None)
            ratio = (avg / 1e3) / (e * 1e3) if e else None
            o.write(f"{n},{cps:.2f},{avg / 1e3:.2f},{'' if e is None else f'{e * 1e3:.2f}'},{'' if ratio is None else f'{ratio:.3f}'}\n")
            if ratio is not None and avg > 20000 and abs(ratio - 1.0) > 0.05:
                CHECK_FAILED = True
        o.write(f"# sum of rocprof averages per step: {tot / 1e3:.1f} us; ms_per_step of the profiled run: {prof_ms}; "
                f"ms_per_step of the un-profiled run: {bl.get('ms_per_step')}\n")
        if prof_ms and tot / 1e6 > prof_ms * 1.001:
            CHECK_FAILED = True
            o.write("# CHECK FAILED: the kernel durations add up to more than the step\n")
    print(open(os.path.join(DST, f"{TAG}_cfg2_timed_region.csv")).read())
traffic = {}
with open(os.path.join(DST, f"{TAG}_hbm_traffic_pmc.csv"), "w") as o:
    o.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, SEPARATE passes, one workload per pass (commands: {TAG}_*_kernel_stats.csv headers)\n"
            "# values are KiB per dispatch as reported; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads,\n"
            "# so read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B stores.\n"
            "workload,kernel,counter,dispatches,mean_KiB,corrected_MiB\n")
    for w in ("cfg2", "pool", "cfg5"):
        per = collections.defaultdict(float)
        for name, sub, mult in (("FETCH_SIZE", "fetch_" + w, 2.0), ("WRITE_SIZE", "write_" + w, 1.0)):
            for k, v in pmc(sub).items():
                vals = v.get(name, [])
                if vals and k.startswith("k_"):
                    m = sum(vals) / len(vals)
                    o.write(f"{w},{k},{name},{len(vals)},{m:.1f},{m * mult / 1024:.1f}\n")
                    per[k] += m * mult * 1024.0
        for k, b in per.items():
            traffic[f"{w}:{k}"] = b
    # config 3: HBM bytes of the WHOLE step = every kernel's bytes over all its dispatches / the number of steps run (one
    # fc_pathology forward launch, k_gemm_nt2, per step)
    tot3, steps3 = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    for name, sub, mult in (("FETCH_SIZE", "fetch_cfg3", 2.0), ("WRITE_SIZE", "write_cfg3", 1.0)):
        for k, v in pmc(sub).items():
            vals = v.get(name, [])
            tot3[name] += sum(vals) * mult * 1024.0
            if k.startswith("k_gemm_nt2"):
                steps3[name] = len(vals)
    # per-kernel rows of config 3's heavy launches (the fused LayerNorm + pool pairs are HBM-bound streams: their bytes per
    # launch against the algorithmic x + dy in, dx out are what DESIGN 4.4 quotes)
    for name, sub, mult in (("FETCH_SIZE", "fetch_cfg3", 2.0), ("WRITE_SIZE", "write_cfg3", 1.0)):
        for k, v in sorted(pmc(sub).items()):
            vals = v.get(name, [])
            if vals and k.startswith(("k_lnbr_", "k_apool_partial", "k_gemm_nt2", "k_gemm_tn2", "k_gate_fwd2", "k_gate_bwd_d", "k_adam_segs")):
                m = sum(vals) / len(vals)
                o.write(f"cfg3,{k},{name},{len(vals)},{m:.1f},{m * mult / 1024:.1f}\n")
    if steps3["FETCH_SIZE"] and steps3["WRITE_SIZE"]:
        rd, wr = tot3["FETCH_SIZE"] / steps3["FETCH_SIZE"], tot3["WRITE_SIZE"] / steps3["WRITE_SIZE"]
        traffic["cfg3:step"] = rd + wr
        o.write(f"cfg3,WHOLE STEP (all kernels / {steps3['FETCH_SIZE']} steps),FETCH_SIZE x 2 + WRITE_SIZE,,,{(rd + wr) / 2 ** 20:.1f}\n")
        o.write(f"cfg3,WHOLE STEP read,FETCH_SIZE x 2,,,{rd / 2 ** 20:.1f}\n")
        o.write(f"cfg3,WHOLE STEP written,WRITE_SIZE,,,{wr / 2 ** 20:.1f}\n")
# the keys bench.py looks up
alias = {}
for key, b in traffic.items():
    w, k = key.split(":", 1)
    if w == "cfg2" and k.startswith("k_gate_fwd"):
        alias["cfg2_gate_fwd"] = b
    if w == "cfg2" and k.startswith("k_gate_bwd_dw"):
        alias["cfg2_gate_bwd_dw"] = b
    if w == "cfg5" and k.startswith("k_gate_fwd_bf16"):
        alias["cfg5_gate_fwd"] = b
    if w == "cfg5" and k.startswith("k_gate_bwd_dw_bf16"):
        alias["cfg5_gate_bwd_dw"] = b
    if w == "pool" and k.startswith("k_pool_partial"):
        alias["pool_4096x512"] = alias.get("pool_4096x512", 0.0) + b
    if w == "pool" and k.startswith("k_pool_merge"):
        alias["pool_4096x512"] = alias.get("pool_4096x512", 0.0) + b
    if w == "cfg3" and k == "step":
        alias["cfg3_step"] = b
if alias:
    json.dump({"_source": f"profiles/{TAG}_hbm_traffic_pmc.csv (2 x FETCH_SIZE + WRITE_SIZE, bytes per launch)", **alias},
              open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(DST, f"{TAG}_mfma_busy_pmc.csv"), "w") as o:
    o.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU\n"
            "# one pass per workload; means per dispatch.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)\n"
            "# (GRBM_GUI_ACTIVE is summed over the 8 XCDs and reads high on dispatches under ~0.3 ms: MI355X_MICROARCH.md, DVFS section).\n"
            "workload,kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_frac,SQ_LDS_BANK_CONFLICT,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_INSTS_VALU\n")
    for w in ("cfg2", "cfg5"):
        for k, v in pmc("mfma_" + w).items():
            if not k.startswith("k_") or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
                continue
            mean = {c: sum(x) / len(x) for c, x in v.items()}
            gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
            frac = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024) if gui else 0.0
            o.write(f"{w},{k},{len(v['SQ_VALU_MFMA_BUSY_CYCLES'])},{mean['SQ_VALU_MFMA_BUSY_CYCLES']:.0f},{gui:.0f},{frac:.3f},"
                    f"{mean.get('SQ_LDS_BANK_CONFLICT', 0):.0f},{mean.get('SQ_WAVE_CYCLES', 0):.0f},{mean.get('SQ_WAIT_ANY', 0):.0f},"
                    f"{mean.get('SQ_WAIT_INST_ANY', 0):.0f},{mean.get('SQ_INSTS_VALU', 0):.0f}\n")
def steady_stats(w, last, out_name):
    """Per-kernel averages over the LAST `last` dispatches of every kernel of workload w (a run of STEPS steps: the first ones
    execute while the clock ramps)."""
    f = sorted(glob.glob(os.path.join(SRC, "stats_" + w, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not f:
        return
    per = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"])):
        per[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(DST, out_name), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace -- STEPS=60 {WORK[w]}\n# averages over the last {last} dispatches of each kernel (steady clock)\n")
        o.write("kernel,dispatches_total,steady_avg_us\n")
        for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-last:])):
            if len(v) >= last and n.startswith("k_"):
                o.write(f"{n},{len(v)},{sum(v[-last:]) / last / 1e3:.2f}\n")


steady_stats("cfg5", 30, f"{TAG}_cfg5_steady.csv")


def replayed_step_stats(w, marker, steps, out_name):
    """cfg3: per-kernel launches and time of ONE REPLAYED step - the dispatches between consecutive launches of
    `marker` (one per step) over the last `steps` steps of the trace, i.e. the hipGraph replays of the timed loop only; the
    whole-run stats file also averages the eager warm-up / capture passes (more launches per step, torch fills and copies)."""
    f = sorted(glob.glob(os.path.join(SRC, "stats_" + w, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not f:
        return
    rows = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f[0]))]
    rows.sort(key=lambda r: r[1])
    idx = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
    if len(idx) < steps + 1:
        return
    agg = collections.defaultdict(lambda: [0, 0.0])
    use = list(zip(idx[-steps - 1:-1], idx[-steps:]))
    for a, b in use:
        for n, s0, e0 in rows[a:b]:
            agg[n][0] += 1
            agg[n][1] += (e0 - s0) / 1e3
    n = len(use)
    span = (rows[use[-1][1]][1] - rows[use[0][0]][1]) / 1e3 / n
    with open(os.path.join(DST, out_name), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace -- {WORK[w]}\n# ONE replayed step: dispatches between consecutive {marker} launches, "
                f"averaged over the last {n} steps (the timed hipGraph replays; eager warm-up / capture passes excluded)\n")
        o.write(f"# launches per step {sum(v[0] for v in agg.values()) / n:.1f}, kernel time per step {sum(v[1] for v in agg.values()) / n:.1f} us, "
                f"step span {span:.1f} us (under the profiler)\n")
        o.write("kernel,launches_per_step,us_per_step,avg_us\n")
        for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{name},{c / n:.2f},{t / n:.2f},{t / c:.2f}\n")


replayed_step_stats("cfg3", "k_gemm_nt2", 15, f"{TAG}_cfg3_replayed_step.csv")
lines = {}
for key, fn in (("bench", "bench_line.json"), ("bench_driver_flags", "bench_driver_line.json"), ("bench_eval_mode", "bench_eval_line.json"),
                ("ragged_one_bag", "ragged_line.json"),
                ("ragged_fusion", "ragged_fusion_line.json"), ("ragged_fusion_10_prompts", "ragged_fusion_p10_line.json"),
                ("ragged_fusion_learnable_prompts", "ragged_fusion_coop_line.json"),
                ("ragged_fusion_ct_plus_pathology", "ragged_fusion_ct_line.json"),
                ("bench_one_rank_through_rccl", "bench_rccl1_line.json"),
                ("fusion", "fusion_line.json"), ("fusion_10_prompts", "p10_line.json"), ("fusion_coop", "coop_line.json"),
                ("one_bag_4096_hipgraph", "one_bag_line.json")):
    try:
        txt = [l for l in open(os.path.join(SRC, fn)).read().splitlines() if l.startswith("{")][-1]
        lines[key] = json.loads(txt)
    except Exception as e:          # noqa: BLE001
        lines[key] = {"error": str(e)}
json.dump(lines, open(os.path.join(DST, f"{TAG}_bench_line.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(DST)))
if CHECK_FAILED:
    print("CHECK FAILED: HIP-event and rocprofv3 durations of a step kernel differ by more than 5 % (or add up to more than the step)")
    sys.exit(1)
