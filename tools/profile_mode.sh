#!/bin/bash
# Collect the rocprofv3 evidence for one sweep mode of bench.py on the GPU box (run from the repo root):
#   bash tools/profile_mode.sh x3 [workload]      -> gpurun_out/prof/<workload>_<mode>_{kernel_stats.csv,kernel_stats_bench.json,
#                                                    pmc_fetch.csv,pmc_write.csv,pmc_sq.csv,stream_gemm_pmc_summary.json}
# One run with --kernel-trace --stats (per-kernel durations) and one run per counter group (PMC passes are never
# combined with other traces).  Copy the files you want judged into profiles/rNN/.
# Environment: STATS_ONLY=1 = only the kernel-trace pass; TIMELINE_ANCHOR=<kernel> = also write the per-iteration timeline
# (tools/timeline.py) of that pass as <workload>_<mode>_timeline.{txt,json}.
set -e
#   bash tools/profile_mode.sh x3 cfg3 fullsig --x-scale 0.3712345     (tag + extra bench.py arguments: file names get <mode>_<tag>)
MODE=${1:-x3}
WL=${2:-cfg3}
TAG=${3:-}
EXTRA="${@:4}"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
MT=$MODE${TAG:+_$TAG}
RAW=$OUT/raw_${WL}_${MT}
mkdir -p "$RAW"
export TMPDIR=/tmp
cd /tmp
COMMON="--workload $WL --dtype $MODE --no-cpu-baseline --no-other-modes $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/stats" -- python3 "$ROOT/bench.py" $COMMON --steps 20 --warmup 3 > "$OUT/${WL}_${MT}_kernel_stats_bench.json" 2> "$RAW/stats.err"
echo "stats pass done"
if [ -n "$TIMELINE_ANCHOR" ]; then
  python3 "$ROOT/tools/timeline.py" "$RAW/stats" "$TIMELINE_ANCHOR" "$OUT/${WL}_${MT}_timeline.json" > "$OUT/${WL}_${MT}_timeline.txt"
fi
if [ -n "$STATS_ONLY" ]; then
  f=$(find "$RAW/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${WL}_${MT}_kernel_stats.csv"
  rm -rf "$RAW"; exit 0
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/fetch" -- python3 "$ROOT/bench.py" $COMMON --steps 3 --warmup 1 > /dev/null 2> "$RAW/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/write" -- python3 "$ROOT/bench.py" $COMMON --steps 3 --warmup 1 > /dev/null 2> "$RAW/write.err"
echo "WRITE_SIZE pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$RAW/sq" -- python3 "$ROOT/bench.py" $COMMON --steps 3 --warmup 1 > /dev/null 2> "$RAW/sq.err"
echo "SQ pass done"
# L2 hit rate of the sweep (panel re-reads that stay inside the XCD vs cross the fabric); optional: a refused counter name must not lose the other passes
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$RAW/tcc" -- python3 "$ROOT/bench.py" $COMMON --steps 3 --warmup 1 > /dev/null 2> "$RAW/tcc.err" && echo "TCC pass done" || echo "TCC pass failed (ignored)"
cd "$ROOT"
python3 tools/summarize_pmc.py "$RAW" "$OUT" "$WL" "$MT"
rm -rf "$RAW"
