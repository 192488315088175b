set -e
export TIMELINE_ANCHOR=w_update_mfma
export STATS_ONLY=1
bash tools/profile_mode.sh x3 cfg3 shard8 --cells 25000
bash tools/profile_mode.sh x3 cfg3 shard4 --cells 50000
bash tools/profile_mode.sh x3 cfg3 shard2 --cells 100000
bash tools/profile_mode.sh x3 cfg2
bash tools/profile_mode.sh f32 cfg2
bash tools/profile_mode.sh x3 cfg4 share8 --cells 125000
bash tools/profile_mode.sh x3 cfg4 share8_fullsig --cells 125000 --x-scale 0.3712345
bash tools/profile_mode.sh f32 cfg4 share8 --cells 125000
unset STATS_ONLY TIMELINE_ANCHOR
python3 bench.py > gpurun_out/prof/bench_default_cfg3.json 2> gpurun_out/prof/bench_default_cfg3.err
ALPINE_BENCH_REHEARSAL_ONE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/prof/bench_rehearsal_2ranks_one_gpu.json 2> gpurun_out/prof/bench_rehearsal_2ranks_one_gpu.err || true
echo part2 done
