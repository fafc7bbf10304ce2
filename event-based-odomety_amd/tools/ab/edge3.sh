# round 3 A/B no. 3: the shipping edge configuration (256 / 768 lanes eval, 256 / 512 solve) against round 2's library
T=event-based-odomety_amd/tools
D=event-based-odomety_amd
M=$D/libebo_hip.so
B=$D/libebo_hip_base.so
for cfg in "0 256" "2 64" "3 16" "4 4"; do
EBO_LIB_PATH=$B python $T/ab_edge.py $cfg "" 2>/dev/null
EBO_LIB_PATH=$M python $T/ab_edge.py $cfg "" 2>/dev/null
done
EBO_LIB_PATH=$M python $T/time_edge_solve.py 0 256 2>/dev/null
EBO_LIB_PATH=$M python $T/time_edge_solve.py 0 1 2>/dev/null
EBO_LIB_PATH=$M python $T/time_edge_solve.py 2 64 2>/dev/null
EBO_LIB_PATH=$M python $T/time_edge_solve.py 3 16 2>/dev/null
EBO_LIB_PATH=$B python $T/time_edge_solve.py 3 16 2>/dev/null
EBO_LIB_PATH=$M python $T/time_reference_call.py 2>/dev/null | tail -12
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
