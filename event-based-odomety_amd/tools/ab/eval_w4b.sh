# round 3: workgroup / LDS split of k_eval3 at 4 waves per SIMD, second sweep
T=event-based-odomety_amd/tools
for cfg in "3 64" "0 256"; do
python $T/ab_eval.py $cfg "" "EBO_EVAL_BLOCK=128,EBO_LDS_KB=25" "EBO_EVAL_BLOCK=128,EBO_LDS_KB=19" "EBO_EVAL_BLOCK=64,EBO_LDS_KB=9" 2>/dev/null
done
for cfg in "2 256" "4 8"; do
python $T/ab_eval.py $cfg "" "EBO_EVAL_BLOCK=256,EBO_LDS_KB=39" "EBO_EVAL_BLOCK=320,EBO_LDS_KB=52" "EBO_EVAL_BLOCK=512,EBO_LDS_KB=79" 2>/dev/null
done
