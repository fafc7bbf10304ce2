#!/bin/bash
# step 5b of tools/profile_round.sh alone: the N > 1 default workload's kernels on one GPU (kernel stats + HBM traffic)
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PY=python3
S="$PY $R/event-based-odomety_amd/tools/pmc_summary.py"
C4="$PY $R/bench.py --workload c4 --steps 5 --warmup 1 --cpu-seconds 1"
rocprofv3 --kernel-trace --stats -d $O/ks_c4 --output-format csv -- $C4 > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
cp $O/ks_c4/*/*kernel_stats.csv $O/${TAG}_bench_c4_1gpu_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $O/c4f --output-format csv -- $C4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/c4w --output-format csv -- $C4 > /dev/null 2>&1
$S k_solve_independent $O/c4f $O/c4w > $O/${TAG}_pmc_k_solve_independent_c4.txt
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
cat $O/${TAG}_pmc_k_solve_independent_c4.txt; grep "k_solve_independent\|k_count_shard" $O/${TAG}_bench_c4_1gpu_kernel_stats.csv | cut -c1-60,150-260
