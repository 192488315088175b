#!/bin/bash
# Everything that goes under profiles/rNN/, in two gpurun calls (a call is limited to 20 minutes) (run from the repo root; ~15 GPU-minutes):
#   bash tools/collect_profiles.sh          -> gpurun_out/prof/*
# cfg3 in every sweep mode with PMC passes (the headline + BASELINE config 5 + the full-significand leg), then kernel traces
# and per-iteration timelines of the other shapes: one of 8 shards of cfg3 (25 000 cells), cfg2, cfg4's per-GPU share
# (125 000 cells, K = 105) in x3 and f32, and the default bench line.
set -e
bash tools/collect_profiles_part1.sh
bash tools/collect_profiles_part2.sh
echo "collect_profiles done"
