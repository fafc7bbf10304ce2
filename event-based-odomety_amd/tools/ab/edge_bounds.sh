set -x
T=event-based-odomety_amd/tools
B=event-based-odomety_amd/libebo_hip_base.so
V=event-based-odomety_amd/libebo_hip_v1.so
for cfg in "0 256" "2 64" "3 16" "4 4"; do
  EBO_LIB_PATH=$B python $T/ab_edge.py $cfg "" 
  EBO_LIB_PATH=$V python $T/ab_edge.py $cfg "EBO_EDGE_BLOCK=384" "EBO_EDGE_BLOCK=768" "EBO_EDGE_BLOCK=512" "EBO_EDGE_BLOCK=256"
done
EBO_LIB_PATH=$B python $T/time_edge_solve.py 0 256
EBO_LIB_PATH=event-based-odomety_amd/libebo_hip.so python $T/time_edge_solve.py 0 256
EBO_LIB_PATH=$V EBO_EDGE_BLOCK=384 python $T/time_edge_solve.py 0 256
EBO_LIB_PATH=$V EBO_EDGE_BLOCK=384 python $T/time_edge_solve.py 0 1
EBO_LIB_PATH=$B python $T/time_edge_solve.py 2 64
EBO_LIB_PATH=$V EBO_EDGE_BLOCK=768 python $T/time_edge_solve.py 2 64
EBO_LIB_PATH=$V EBO_EDGE_BLOCK=384 python -m pytest tests/test_gpu_edge.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -5
