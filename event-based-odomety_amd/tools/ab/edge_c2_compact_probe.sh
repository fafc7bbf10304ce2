#!/bin/bash
# probe: C2 / C4 (canvas does not fit a CU's LDS twice: one 768-lane workgroup per CU today) forced onto the aliased layout at
# 80 KB, 256 lanes, with and without the compact path -- how much would the three-per-CU path buy there?  (d != 0: another
# lane count reduces the unit's sums in another order)
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
A="EBO_EDGE_LAYOUT=1,EBO_EDGE_LDS_KB=80,EBO_EDGE_BLOCK=256"
python $T/ab_edge.py 2 64 "" "$A,EBO_EDGE_COMPACT=0" "$A,EBO_EDGE_COMPACT=1" "" 2>&1 | grep -v amdgpu.ids
python $T/ab_edge.py 4 8 "" "$A,EBO_EDGE_COMPACT=0" "$A,EBO_EDGE_COMPACT=1" "" 2>&1 | grep -v amdgpu.ids
python $T/ab_edge.py 2 4 "" "$A,EBO_EDGE_COMPACT=0" "" 2>&1 | grep -v amdgpu.ids
python $T/ab_edge.py 2 1 "" "$A,EBO_EDGE_COMPACT=0" "" 2>&1 | grep -v amdgpu.ids
