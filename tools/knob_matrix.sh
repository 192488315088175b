# The parity suites under every result-preserving A/B knob of the production library (one setting at a time):
#   gpurun -- bash tools/knob_matrix.sh [first|second]      (two halves: a gpurun call is limited to 20 minutes)
set -e
A="ALPINE_HIP_X3_VARIANT=2 ALPINE_HIP_X3_VARIANT=0 ALPINE_HIP_NO_TAIL=1 ALPINE_HIP_FUSED_W=0 ALPINE_HIP_UNFUSED_MID=1 ALPINE_HIP_H_UPDATE=valu ALPINE_HIP_X3_NARROW=0 ALPINE_HIP_BF16_WAVES=4"
B="ALPINE_HIP_SG_VARIANT=1 ALPINE_HIP_SG_VARIANT=2 ALPINE_HIP_GUIDED=scalar ALPINE_HIP_TAIL_STATS=per_covariate ALPINE_HIP_XCD_BIAS=0 ALPINE_HIP_XCD_BIAS=40"
case "${1:-all}" in first) LIST="$A";; second) LIST="$B";; *) LIST="$A $B";; esac
for kv in $LIST; do
  echo "== $kv"
  env $kv timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1
done
