#!/bin/bash
# The round-5 A/B of the edge loss's compact layout (three workgroups per CU + deferred launch) against the single launch
# of rounds 1-4, by LDS budget and flow scale, one process per flow scale (profiles/r05_edge_levers.txt, item 7):
#   gpurun -- bash event-based-odomety_amd/tools/ab/edge_compact_ab.sh
set -e
cd ${GRAFT_REPO_ROOT:-.}
for fs in 0.0 0.5 1.0; do
  export EBO_AB_FLOWSCALE=$fs
  echo "flow scale $fs"
  python event-based-odomety_amd/tools/ab_edge.py 0 256 "EBO_EDGE_COMPACT=0" "" "EBO_EDGE_COMPACT_KB=46" "EBO_EDGE_COMPACT_KB=48" \
      "EBO_EDGE_COMPACT_KB=51" "EBO_EDGE_COMPACT_KB=52" "EBO_EDGE_COMPACT_KB=53" "EBO_EDGE_COMPACT=0"
done
export EBO_AB_FLOWSCALE=0.5
python event-based-odomety_amd/tools/ab_edge.py 3 64 "EBO_EDGE_COMPACT=0" "" "EBO_EDGE_COMPACT_KB=52"
export EBO_LIB_PATH=$PWD/event-based-odomety_amd/libebo_hip_ab.so
echo "lock-step reference-default call, compact"; python event-based-odomety_amd/tools/time_reference_call.py 0 1 16 64 256
echo "lock-step reference-default call, EBO_EDGE_COMPACT=0"; EBO_EDGE_COMPACT=0 python event-based-odomety_amd/tools/time_reference_call.py 0 1 16 64 256
