# round 3: k_solve_independent at three waves per SIMD (152-161 VGPRs: one next_step() site, solver state volatile in LDS);
# base = the library before the change (205 VGPRs, two waves, 128 lanes x 40 KB)
T=event-based-odomety_amd/tools
D=event-based-odomety_amd
for cfg in "3 64" "0 256"; do
EBO_LIB_PATH=$D/libebo_hip_base.so python $T/ab_solve.py $cfg "" 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip.so python $T/ab_solve.py $cfg "" "EBO_LDS_KB=26" "EBO_LDS_KB=22" 2>/dev/null
done
for cfg in "2 64" "4 4"; do
EBO_LIB_PATH=$D/libebo_hip_base.so python $T/ab_solve.py $cfg "" 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip.so python $T/ab_solve.py $cfg "" "EBO_LDS_KB=26" "EBO_LDS_KB=39,EBO_SOLVE_BLOCK=192" "EBO_LDS_KB=52,EBO_SOLVE_BLOCK=256" "EBO_LDS_KB=31,EBO_SOLVE_BLOCK=128" 2>/dev/null
done
