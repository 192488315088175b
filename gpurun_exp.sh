mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
ALPINE_HIP_H_UPDATE=valu timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu_valu.log 2>&1; tail -2 gpurun_out/pytest_gpu_valu.log
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/exp_$tag.log 2>&1; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp_$tag.log").read().strip().splitlines()[-1])
    r=d["roofline"]; print("$tag", "it/s=%.1f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], "TF=%.1f"%r["achieved"], "xht=%.3f wtx=%.3f ms"%(r["avg_ms_xht"], r["avg_ms_wtx"]), "share=%.3f"%r["sweeps_share_of_step"])
except Exception as e: print("$tag FAILED", e)
PY
}
run mfma
ALPINE_HIP_H_UPDATE=valu run valu
run mfma2
