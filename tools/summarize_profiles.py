"""Turn gpurun_out/prof_round/ (tools/profile_round.sh) into the small tracked summaries under profiles/."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def stats(sub, out, header, top=40):
    f = glob.glob(os.path.join(SRC, sub, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(DST, out), "w") as o:
        o.write(header)
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:top]:
            o.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.1f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")


def pmc(sub):
    f = glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


os.makedirs(DST, exist_ok=True)
stats("stats", f"{TAG}_bench_kernel_stats.csv",
      "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline   (MI355X, 1 GPU)\n"
      "# the run also contains bench.py's per-kernel breakdown pass and the N=4096 pool-roofline pass (pool kernel averages\n"
      "# mix the N=1024 and N=4096 shapes).  Durations under the profiler read a few % above bench.py's HIP-event times.\n")
stats("fusion", f"{TAG}_fusion_kernel_stats.csv",
      "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_fusion.py --cache_text --steps 20 --warmup 5\n"
      "# BASELINE config 3 (32 bags x 1024 x 768 + CLIP ViT-B/32 text): 25 eager steps; the first step also runs the frozen text tower\n", top=60)
stats("coop", f"{TAG}_coop_kernel_stats.csv",
      "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_fusion.py --coop --steps 6 --warmup 2\n"
      "# upstream's default mode: learnable prompts (10 per bag) trained THROUGH the frozen ViT-B/32 text tower, 32 bags x 1024 x 768\n", top=30)
stats("bf16", f"{TAG}_bf16_kernel_stats.csv",
      "# rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype bf16 --patches 4096 --dim 1024 --steps 30 --warmup 5\n"
      "# BASELINE config 5 (32 bags x 4096 x 1024, bf16 storage)\n", top=20)
fetch, write = pmc("fetch"), pmc("write")
with open(os.path.join(DST, f"{TAG}_bench_hbm_traffic_pmc.csv"), "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown\n"
            "# values are KiB per dispatch as reported; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads,\n"
            "# so read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B stores.\n"
            "kernel,counter,dispatches,mean_KiB,corrected_MiB\n")
    for name, agg, mult in (("FETCH_SIZE", fetch, 2.0), ("WRITE_SIZE", write, 1.0)):
        for k, v in agg.items():
            vals = v.get(name, [])
            if vals and k.startswith("k_"):
                m = sum(vals) / len(vals)
                o.write(f"{k},{name},{len(vals)},{m:.1f},{m * mult / 1024:.1f}\n")
mf = pmc("mfma")
with open(os.path.join(DST, f"{TAG}_bench_mfma_busy_pmc.csv"), "w") as o:
    o.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY\n"
            "# -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown; means per dispatch.\n"
            "# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs\n"
            "# (MI355X_MICROARCH.md, DVFS section), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs.\n"
            "kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_frac,SQ_LDS_BANK_CONFLICT,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY\n")
    for k, v in mf.items():
        if not k.startswith("k_") or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
            continue
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
        frac = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024) if gui else 0.0
        o.write(f"{k},{len(v['SQ_VALU_MFMA_BUSY_CYCLES'])},{mean['SQ_VALU_MFMA_BUSY_CYCLES']:.0f},{gui:.0f},{frac:.3f},"
                f"{mean.get('SQ_LDS_BANK_CONFLICT', 0):.0f},{mean.get('SQ_WAVE_CYCLES', 0):.0f},{mean.get('SQ_WAIT_ANY', 0):.0f},{mean.get('SQ_WAIT_INST_ANY', 0):.0f}\n")
lines = {}
for key, fn in (("bench", "bench_line.json"), ("fusion", "fusion_line.json"), ("bf16", "bf16_line.json"),
                ("fusion_coop", "coop_line.json"), ("fusion_10_prompts", "p10_line.json"),
                ("fusion_coop_split3", "coop3_line.json"), ("fusion_coop_split2", "coop2_line.json"),
                ("one_bag_4096_hipgraph", "one_bag_line.json"), ("one_bag_4096_10_prompts_hipgraph", "one_bag_p10_line.json"),
                ("one_bag_4096_learnable_prompts_hipgraph", "one_bag_coop_line.json")):
    try:
        txt = [l for l in open(os.path.join(SRC, fn)).read().splitlines() if l.startswith("{")][-1]
        lines[key] = json.loads(txt)
    except Exception as e:          # noqa: BLE001
        lines[key] = {"error": str(e)}
json.dump(lines, open(os.path.join(DST, f"{TAG}_bench_line.json"), "w"), indent=1)
try:
    txt = open(os.path.join(SRC, "kbench_split.txt")).read()
    open(os.path.join(DST, f"{TAG}_split_gemm_kbench.txt"), "w").write(
        "# python3 tools/kbench_split.py: fp32 MFMA GEMM vs split-bf16 products (2 / 3 pieces), error against float64 on 512 rows\n" + txt)
except OSError:
    pass
print("wrote", sorted(os.listdir(DST)))
