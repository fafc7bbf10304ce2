set -e
cd $GRAFT_REPO_ROOT
python event-based-odomety_amd/tools/ab_edge.py 0 256 "EBO_EDGE_ABLATE=512" "" "EBO_EDGE_ABLATE=512" ""
python event-based-odomety_amd/tools/ab_edge.py 3 64 "EBO_EDGE_ABLATE=512" "" "EBO_EDGE_ABLATE=512" ""
python event-based-odomety_amd/tools/ab_edge.py 2 64 "EBO_EDGE_ABLATE=512" "" "EBO_EDGE_ABLATE=512" ""
python event-based-odomety_amd/tools/ab_edge.py 4 8 "EBO_EDGE_ABLATE=512" "" "EBO_EDGE_ABLATE=512" ""
