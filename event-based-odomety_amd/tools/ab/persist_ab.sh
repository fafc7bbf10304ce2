set -e
cd $GRAFT_REPO_ROOT
python event-based-odomety_amd/tools/ab_edge.py 0 256 "" "EBO_EDGE_ABLATE=256" "" "EBO_EDGE_ABLATE=256" "EBO_EDGE_COMPACT=0" "EBO_EDGE_COMPACT=0,EBO_EDGE_ABLATE=256"
python event-based-odomety_amd/tools/ab_edge.py 3 64 "" "EBO_EDGE_ABLATE=256" "" "EBO_EDGE_ABLATE=256"
