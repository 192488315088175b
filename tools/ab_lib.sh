# Two builds of the library on ONE box, alternating processes (each process allocates afresh: +-2 % placement noise per run, hence
# the repeats): bash tools/ab_lib.sh <old.so> <new.so> [repeats] [extra bench.py arguments]
OLD=$1; NEW=$2; REP=${3:-3}; shift 3 || true
for i in $(seq 1 $REP); do
  for lib in "$OLD" "$NEW"; do
    ALPINE_HIP_LIBRARY=$lib python3 bench.py --no-cpu-baseline --no-other-modes --steps 50 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms; sweep', round(d['roofline']['avg_launch_ms'],4), 'ms', d['roofline']['kernel_name'])"
  done
done
