# round 3 A/B no. 4: the lane's first event kept in registers across the edge kernel's passes
T=event-based-odomety_amd/tools
D=event-based-odomety_amd
for cfg in "0 256" "2 64" "3 16" "4 4"; do
EBO_LIB_PATH=$D/libebo_hip_base.so python $T/ab_edge.py $cfg "" 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip.so python $T/ab_edge.py $cfg "" 2>/dev/null
done
EBO_LIB_PATH=$D/libebo_hip_base.so python $T/time_edge_solve.py 0 256 2>/dev/null
EBO_LIB_PATH=$D/libebo_hip.so python $T/time_edge_solve.py 0 256 2>/dev/null
