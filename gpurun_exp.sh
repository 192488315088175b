mkdir -p gpurun_out/r01b && export TMPDIR=/tmp
P=gpurun_out/r01b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_f32 -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $P/stats_f32.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_bf16 -- python3 bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > $P/stats_bf16.log 2>&1 &&
for dt in f32 bf16; do
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch_$dt -- python3 bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $P/fetch_$dt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write_$dt -- python3 bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $P/write_$dt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/sq_$dt -- python3 bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $P/sq_$dt.log 2>&1
done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $P/bench_default.log 2>&1
tail -c 600 $P/bench_default.log
find $P -name "*.csv" | wc -l
