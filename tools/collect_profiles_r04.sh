#!/bin/bash
# Round 4: the rocprofv3 evidence kept under profiles/r04/ (run from the repo root on the GPU box; two gpurun calls: part a / part b).
#   bash tools/collect_profiles_r04.sh a|b     -> gpurun_out/prof/*
# a: PMC passes (FETCH_SIZE, WRITE_SIZE, SQ group, L2 hit rate) + kernel stats of the headline (cfg3), of cfg4's per-GPU share (K = 105:
#    the shape VERDICT r3 asked to profile) on count data and on full significands, and of the K = 150 one-pass sweep
# b: kernel traces + per-iteration timelines of the shard / cfg2 shapes, the default bench line, the 2-rank rehearsal line
set -e
PART=${1:-a}
export TIMELINE_ANCHOR=w_update_mfma
if [ "$PART" = a ]; then
  bash tools/profile_mode.sh x3 cfg3
  bash tools/profile_mode.sh x3 cfg3 fullsig --x-scale 0.3712345
  bash tools/profile_mode.sh x3 cfg4 share8 --cells 125000
  bash tools/profile_mode.sh x3 cfg4 share8_fullsig --cells 125000 --x-scale 0.3712345
  TIMELINE_ANCHOR=wide_w_apply bash tools/profile_mode.sh x3 cfg3_k150
  echo part a done
else
  export STATS_ONLY=1
  bash tools/profile_mode.sh x3 cfg3 shard8 --cells 25000
  bash tools/profile_mode.sh x3 cfg3 shard4 --cells 50000
  bash tools/profile_mode.sh x3 cfg3 shard2 --cells 100000
  bash tools/profile_mode.sh x3 cfg2
  bash tools/profile_mode.sh f32 cfg3
  bash tools/profile_mode.sh bf16 cfg3
  unset STATS_ONLY TIMELINE_ANCHOR
  python3 bench.py > gpurun_out/prof/bench_default_cfg3.json 2> gpurun_out/prof/bench_default_cfg3.err
  python3 bench.py --workload cfg4 --cells 125000 --no-cpu-baseline > gpurun_out/prof/bench_cfg4_share8.json 2> gpurun_out/prof/bench_cfg4_share8.err
  python3 bench.py --cells 25000 --steps 200 --warmup 20 --no-cpu-baseline --no-other-modes > gpurun_out/prof/bench_cfg3_shard8.json 2> gpurun_out/prof/bench_cfg3_shard8.err
  python3 bench.py --workload cfg3_k150 --no-cpu-baseline > gpurun_out/prof/bench_wide_k150.json 2> gpurun_out/prof/bench_wide_k150.err
  python3 bench.py --workload cfg3_k256 --no-cpu-baseline --no-other-modes > gpurun_out/prof/bench_wide_k256.json 2> gpurun_out/prof/bench_wide_k256.err
  ALPINE_BENCH_REHEARSAL_ONE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/prof/bench_rehearsal_2ranks_one_gpu.json 2> gpurun_out/prof/bench_rehearsal_2ranks_one_gpu.err || true
  echo part b done
fi
