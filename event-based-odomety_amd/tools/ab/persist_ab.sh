set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_edge.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -5
B="EBO_EDGE_ABLATE=256"
for fs in 0.0 0.5 1.0; do
export EBO_AB_FLOWSCALE=$fs
echo "flow scale $fs"
python event-based-odomety_amd/tools/ab_edge.py 0 256 "$B,EBO_EDGE_COMPACT=0" "$B" "$B,EBO_EDGE_COMPACT_KB=46" "$B,EBO_EDGE_COMPACT_KB=48" "$B,EBO_EDGE_COMPACT_KB=51" "$B,EBO_EDGE_COMPACT_KB=52" "$B,EBO_EDGE_COMPACT_KB=53" "$B,EBO_EDGE_COMPACT=0"
done
export EBO_AB_FLOWSCALE=0.5
python event-based-odomety_amd/tools/ab_edge.py 3 64 "$B,EBO_EDGE_COMPACT=0" "$B" "$B,EBO_EDGE_COMPACT_KB=52"
