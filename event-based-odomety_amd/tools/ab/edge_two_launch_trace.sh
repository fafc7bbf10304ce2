#!/bin/bash
# how long the two launches of the compact path take (reference default x 256, the bench's flows): kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/two_launch
mkdir -p $OUT
for fs in 0.5 1.0; do
EBO_AB_FLOWSCALE=$fs rocprofv3 --kernel-trace --stats -d $OUT/fs$fs -o t --output-format csv -- python3 $R/event-based-odomety_amd/tools/ab_edge.py 0 256 "" > $OUT/run$fs.txt 2>&1
grep -h "k_eval_edge" $OUT/fs$fs/t_kernel_stats.csv | cut -c1-60,300-
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/fs$fs/t_kernel_stats.csv"))]
for r in rows[:6]:
    print(r["Name"][:70], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3))
PY
done
