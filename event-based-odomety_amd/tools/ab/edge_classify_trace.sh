#!/bin/bash
# kernel durations of the compact path with / without k_edge_classify, flows 0.5 x and 1.0 x the ground truth
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/classify_trace
mkdir -p $OUT
for fs in 0.5 1.0; do
for st in "EBO_EDGE_CLASSIFY=1" "EBO_EDGE_CLASSIFY=0"; do
echo "== flows $fs x, $st"
EBO_AB_FLOWSCALE=$fs rocprofv3 --kernel-trace --stats -d $OUT/fs$fs$st -o t --output-format csv -- python3 $R/event-based-odomety_amd/tools/ab_edge.py 0 256 "$st" > $OUT/run.txt 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/fs$fs$st/t_kernel_stats.csv"))]
for r in rows[:7]:
    print("   ", r["Name"][:74], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3))
PY
done
done
