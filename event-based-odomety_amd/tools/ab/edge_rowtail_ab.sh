#!/bin/bash
# A/B: single-row tasks for the last, mostly empty round of the reverse sweep in the rolled (168-VGPR) instantiation too
set -e
cd ${GRAFT_REPO_ROOT:-.}
python event-based-odomety_amd/tools/ab_edge.py 0 256 "" "EBO_EDGE_ABLATE=1024" "" "EBO_EDGE_ABLATE=1024"
python event-based-odomety_amd/tools/ab_edge.py 3 64 "" "EBO_EDGE_ABLATE=1024" "" "EBO_EDGE_ABLATE=1024"
python event-based-odomety_amd/tools/ab_edge.py 2 64 "" "EBO_EDGE_ABLATE=1024" "" "EBO_EDGE_ABLATE=1024"
python event-based-odomety_amd/tools/ab_edge.py 4 8 "" "EBO_EDGE_ABLATE=1024" "" "EBO_EDGE_ABLATE=1024"
