set -e
export TIMELINE_ANCHOR=w_update_mfma
bash tools/profile_mode.sh x3 cfg3
bash tools/profile_mode.sh x3 cfg3 fullsig --x-scale 0.3712345
bash tools/profile_mode.sh f32 cfg3
bash tools/profile_mode.sh split cfg3
bash tools/profile_mode.sh bf16 cfg3
echo part1 done
