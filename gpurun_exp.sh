mkdir -p gpurun_out
S=$(date +%s); timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "default bench wall $(( $(date +%s) - S )) s"
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_default.log") if l.startswith("{")][-1])
print("headline", d["dtype"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k,v in (d["other_modes"] or {}).items(): print(k, v.get("value"), v.get("roofline",{}).get("bound"), v.get("roofline",{}).get("frac"), v.get("final_total_loss_rel_diff_vs_headline"), v.get("error"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
ALPINE_BENCH_REHEARSAL_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --cells 60000 --steps 5 --warmup 1 > gpurun_out/bench_2rank.log 2>&1; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_2rank.log') if l.startswith('{')][-1]); print(d['n_gpus'], d['value'], d['final_loss_row'][:2], d['other_modes'])"
