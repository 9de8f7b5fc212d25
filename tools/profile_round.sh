#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun).  One run PER SHAPE, so that
# every per-kernel average in profiles/ belongs to one workload:
#   cfg2   bench.py default step (32 x 1024 x 512 fp32, train mode), step kernels only (--no-breakdown --no-configs), the
#          bench's default priming; tools/summarize_profiles.py averages the dispatches of the TIMED region only
#   pool   the attention-pool stage alone at 64 x 4096 x 512 (tools/prof_pool.py)
#   cfg5   32 x 4096 x 1024 bf16 step (tools/prof_stage.py --bf16)
#   cfg3   32 x 1024 x 768 fusion step (tools/bench_fusion.py --graph)
# kernel-trace stats, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA-busy set).  Raw output goes to
# gpurun_out/prof_round/; tools/summarize_profiles.py writes the tracked summaries.  rocprofv3 gets the python program
# itself after "--" (no wrappers).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_round
# PART=A: the profiled runs (kernel traces + PMC passes); PART=B: the un-profiled bench.py lines; PART=C: the tools/ lines;
# unset: all (a gpurun call is limited to 20 minutes - run the parts as separate calls)
PART=${PART:-ABC}
# ONLY="cfg3" (with PART=A): re-profile just these workloads, keeping the other raw runs of the round
WL=${ONLY:-cfg2 pool cfg5 cfg3}
if [[ "$PART" == *A* && -z "$ONLY" ]]; then rm -rf "$OUT"; fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-configs --no-cpu-baseline --no-breakdown"
declare -A CMD
CMD[cfg2]="$BENCH --regions 3 --steps 200 --warmup 10"      # the bench's own priming (--prime 300): the chip at its steady clock
CMD[pool]="python3 $ROOT/tools/prof_pool.py"
CMD[cfg5]="python3 $ROOT/tools/prof_stage.py --bf16"           # STEPS=60 below: the last 30 steps are the steady state
CMD[cfg3]="python3 $ROOT/tools/bench_fusion.py --graph --steps 20 --warmup 3"
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
# QUICK=1: the cfg2 kernel trace + one un-profiled bench line only (the HIP-event vs rocprofv3 cross-check)
if [ "${QUICK:-0}" = "1" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2 -- ${CMD[cfg2]} > $OUT/stats_cfg2.log 2>&1
  python3 $ROOT/bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench_line.err
  find $OUT -name "*agent_info*" -delete
  echo quick done
  exit 0
fi
export STEPS=60
if [[ "$PART" == *A* ]]; then
for w in $WL; do
  rm -rf $OUT/stats_$w $OUT/fetch_$w $OUT/write_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- ${CMD[$w]} > $OUT/stats_$w.log 2>&1
  echo "stats $w done"
done
for w in $WL; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$w -- ${CMD[$w]} > $OUT/fetch_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$w -- ${CMD[$w]} > $OUT/write_$w.log 2>&1
  echo "traffic $w done"
done
for w in cfg2 cfg5; do
  [[ " $WL " == *" $w "* ]] || continue
  timeout -k 10 300 rocprofv3 --pmc $MFMA --output-format csv -d $OUT/mfma_$w -- ${CMD[$w]} > $OUT/mfma_$w.log 2>&1
  echo "mfma $w done"
done
find $OUT -name "*agent_info*" -delete
fi
if [[ "$PART" == *B* ]]; then
python3 $ROOT/bench.py --steps 200 --warmup 20 > $OUT/bench_line.json 2> $OUT/bench_line.err
echo "bench done"
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2>/dev/null      # the driver's flags
python3 $ROOT/bench.py --steps 200 --warmup 20 --train-mode 0 --no-configs --no-cpu-baseline --no-rccl-floor > $OUT/bench_eval_line.json 2>/dev/null
MIL_FORCE_COLLECTIVES=1 python3 $ROOT/bench.py --gpus 1 --no-configs --no-cpu-baseline --no-rccl-floor > $OUT/bench_rccl1_line.json 2>/dev/null
echo "part B done"
fi
if [[ "$PART" != *C* ]]; then exit 0; fi
python3 $ROOT/tools/bench_ragged.py > $OUT/ragged_line.json 2>/dev/null
python3 $ROOT/tools/bench_ragged.py --fusion > $OUT/ragged_fusion_line.json 2>/dev/null
python3 $ROOT/tools/bench_ragged.py --fusion --prompts10 > $OUT/ragged_fusion_p10_line.json 2>/dev/null
python3 $ROOT/tools/bench_ragged.py --fusion --coop > $OUT/ragged_fusion_coop_line.json 2>/dev/null
python3 $ROOT/tools/bench_ragged.py --fusion --ct > $OUT/ragged_fusion_ct_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --steps 50 --warmup 5 > $OUT/fusion_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --prompts 10 --steps 20 --warmup 3 > $OUT/p10_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --coop --steps 10 --warmup 3 > $OUT/coop_line.json 2>/dev/null
python3 $ROOT/tools/bench_fusion.py --graph --bags 1 --patches 4096 --steps 50 --warmup 5 > $OUT/one_bag_line.json 2>/dev/null
echo done
