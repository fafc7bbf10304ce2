set -e
cd $GRAFT_REPO_ROOT
python event-based-odomety_amd/tools/ab_count.py 3 512 "EBO_COUNT_ALTERNATE=0" "" "EBO_COUNT_ALTERNATE=0" ""
python event-based-odomety_amd/tools/ab_count.py 4 72 "EBO_COUNT_ALTERNATE=0" "" "EBO_COUNT_ALTERNATE=0" ""
python event-based-odomety_amd/tools/ab_count.py 2 1536 "EBO_COUNT_ALTERNATE=0" "" 
