# round 3: k_eval3 at 4 waves per SIMD (128 VGPRs, sums in per-wave LDS slots) against round 3's earlier library
# (libebo_hip_base.so = 137 VGPRs, 3 waves, 40 KB); then the workgroup / LDS split at 4 waves
T=event-based-odomety_amd/tools
D=event-based-odomety_amd
for cfg in "3 64" "2 256" "4 8" "0 256"; do
echo "== base"; EBO_LIB_PATH=$D/libebo_hip_base.so python $T/ab_eval.py $cfg "" 2>/dev/null
echo "== new"; EBO_LIB_PATH=$D/libebo_hip.so python $T/ab_eval.py $cfg "" "EBO_LDS_KB=40" "EBO_EVAL_BLOCK=256,EBO_LDS_KB=39" "EBO_EVAL_BLOCK=128,EBO_LDS_KB=19" "EBO_EVAL_BLOCK=128,EBO_LDS_KB=22" 2>/dev/null
done
