mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for i in 1 2; do timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/exp_b$i.log 2>&1; python - <<PY
import json
d=json.loads(open("gpurun_out/exp_b$i.log").read().strip().splitlines()[-1])
r=d["roofline"]; print("it/s=%.1f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], "TF=%.1f"%r["achieved"], "xht=%.3f wtx=%.3f ms"%(r["avg_ms_xht"], r["avg_ms_wtx"]), "share=%.3f"%r["sweeps_share_of_step"])
PY
done
