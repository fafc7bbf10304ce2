set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r04c4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EBO_LIB_PATH=$R/event-based-odomety_amd/libebo_hip.so
C4="python3 $R/bench.py --workload c4 --steps 3 --warmup 1 --preheat 0 --cpu-seconds 0.5"
rocprofv3 --kernel-trace --stats -d $O/ks_c4 --output-format csv -- $C4 > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
cp $O/ks_c4/*/*kernel_stats.csv $O/r04_bench_c4_1gpu_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $O/c4f --output-format csv -- $C4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/c4w --output-format csv -- $C4 > /dev/null 2>&1
python3 $R/event-based-odomety_amd/tools/pmc_summary.py k_solve_independent $O/c4f $O/c4w > $O/r04_pmc_k_solve_independent_c4.txt
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
cat $O/r04_pmc_k_solve_independent_c4.txt; head -3 $O/r04_bench_c4_1gpu_kernel_stats.csv | cut -c1-200
