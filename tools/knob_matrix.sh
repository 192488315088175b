# The parity suites under every result-preserving A/B knob (one setting at a time).  Round 4: the PRODUCTION library no longer reads
# these variables (explicit alpine_debug_set_option calls instead, exercised by tests/test_gpu_parity.py); the DIAGNOSTICS build does,
# so the matrix runs against it:  python alpine_amd/build.py --diag, then
#   gpurun -- bash tools/knob_matrix.sh [first|second]      (two halves: a gpurun call is limited to 20 minutes)
set -e
export ALPINE_HIP_LIBRARY=${ALPINE_HIP_LIBRARY:-alpine_amd/libalpine_hip_diag.so}
A="ALPINE_HIP_X3_VARIANT=2 ALPINE_HIP_X3_VARIANT=0 ALPINE_HIP_NO_TAIL=1 ALPINE_HIP_FUSED_W=0 ALPINE_HIP_UNFUSED_MID=1 ALPINE_HIP_X3_NARROW=0 ALPINE_HIP_BF16_WAVES=4"
B="ALPINE_HIP_SG_VARIANT=1 ALPINE_HIP_SG_VARIANT=2 ALPINE_HIP_GUIDED=scalar ALPINE_HIP_TAIL_STATS=per_covariate ALPINE_HIP_XCD_BIAS=0 ALPINE_HIP_XCD_BIAS=40"
case "${1:-all}" in first) LIST="$A";; second) LIST="$B";; *) LIST="$A $B";; esac
for kv in $LIST; do
  echo "== $kv"
  env $kv timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1
done
