#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   kernel-trace stats of the default bench.py run, separate PMC passes (HBM traffic, MFMA busy), the fusion
#   (config 3) and bf16 (config 5) kernel stats.  Raw output goes to gpurun_out/prof_round/, summaries are written
#   by tools/summarize_profiles.py.  rocprofv3 gets the python program itself after "--" (no wrappers).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 50 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown > $OUT/mfma.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fusion -- python3 $ROOT/tools/bench_fusion.py --cache_text --steps 20 --warmup 5 > $OUT/fusion.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/coop -- python3 $ROOT/tools/bench_fusion.py --coop --steps 6 --warmup 2 > $OUT/coop.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16 -- $B --dtype bf16 --patches 4096 --dim 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > $OUT/bf16.log 2>&1
tail -1 $OUT/stats.log > $OUT/bench_line_under_profiler.json
python3 $ROOT/bench.py --steps 200 --warmup 20 > $OUT/bench_line.json 2> $OUT/bench_line.err
python3 $ROOT/tools/bench_fusion.py --graph --steps 50 --warmup 5 > $OUT/fusion_line.json 2>/dev/null
python3 $ROOT/bench.py --dtype bf16 --patches 4096 --dim 1024 --no-cpu-baseline --no-breakdown > $OUT/bf16_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --coop --steps 10 --warmup 3 > $OUT/coop_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --prompts 10 --steps 20 --warmup 3 > $OUT/p10_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --coop --clip_gemm_pieces 3 --steps 10 --warmup 3 > $OUT/coop3_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --coop --clip_gemm_pieces 2 --steps 10 --warmup 3 > $OUT/coop2_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --bags 1 --patches 4096 --steps 50 --warmup 5 > $OUT/one_bag_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --bags 1 --patches 4096 --prompts 10 --steps 50 --warmup 5 > $OUT/one_bag_p10_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --bags 1 --patches 4096 --coop --steps 30 --warmup 5 > $OUT/one_bag_coop_line.json 2>/dev/null
python3 $ROOT/tools/kbench_split.py > $OUT/kbench_split.txt 2>/dev/null
echo done
