# round 3 A/B no. 2: per-MAXT edge kernels, reverse-pass unroll at 256 VGPRs, split solve, one-lane LM for the variance solve
T=event-based-odomety_amd/tools
D=event-based-odomety_amd
M=$D/libebo_hip.so
B=$D/libebo_hip_base.so
echo "== main lib"
EBO_LIB_PATH=$M python $T/ab_edge.py 0 256 "" "EBO_EDGE_BLOCK=256" 2>/dev/null
EBO_LIB_PATH=$M python $T/ab_edge.py 2 64 "" "EBO_EDGE_BLOCK=768" 2>/dev/null
EBO_LIB_PATH=$M python $T/ab_edge.py 3 16 "" "EBO_EDGE_BLOCK=256" 2>/dev/null
EBO_LIB_PATH=$M python $T/ab_edge.py 4 4 "" "EBO_EDGE_BLOCK=768" 2>/dev/null
echo "== v3 (reverse pass fully unrolled)"
EBO_LIB_PATH=$D/libebo_hip_v3.so python $T/ab_edge.py 0 256 "EBO_EDGE_BLOCK=256" 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip_v3.so python $T/ab_edge.py 3 16 "EBO_EDGE_BLOCK=256" 2>/dev/null
echo "== edge solve: main / main@256 / v4 (split) / v4@256"
EBO_LIB_PATH=$M python $T/time_edge_solve.py 0 256 2>/dev/null
EBO_LIB_PATH=$M EBO_EDGE_BLOCK=256 python $T/time_edge_solve.py 0 256 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip_v4.so python $T/time_edge_solve.py 0 256 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip_v4.so EBO_EDGE_BLOCK=256 python $T/time_edge_solve.py 0 256 2>/dev/null
EBO_LIB_PATH=$M python $T/time_edge_solve.py 0 1 2>/dev/null
EBO_LIB_PATH=$M EBO_EDGE_BLOCK=768 python $T/time_edge_solve.py 2 64 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip_v4.so EBO_EDGE_BLOCK=768 python $T/time_edge_solve.py 2 64 2>/dev/null
echo "== variance solve: base / main"
EBO_LIB_PATH=$B python $T/ab_solve.py 3 64 "" "EBO_SOLVE_BLOCK=192" "EBO_SOLVE_BLOCK=256" 2>/dev/null
EBO_LIB_PATH=$M python $T/ab_solve.py 3 64 "" "EBO_SOLVE_BLOCK=192" "EBO_SOLVE_BLOCK=256" 2>/dev/null
EBO_LIB_PATH=$B python $T/ab_solve.py 4 4 "" 2>/dev/null
EBO_LIB_PATH=$M python $T/ab_solve.py 4 4 "" "EBO_SOLVE_BLOCK=192" 2>/dev/null
echo "== tests (main lib)"
python -m pytest tests/test_gpu_edge.py tests/test_gpu_random.py tests/test_gpu_shard.py tests/test_gpu_multiprocess.py tests/test_gpu_parity.py tests/test_tracks.py tests/test_gpu_comm.py -x -q -m gpu 2>&1 | tail -8
