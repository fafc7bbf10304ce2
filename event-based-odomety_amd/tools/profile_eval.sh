#!/bin/bash
# kernel stats + counters of the N = 1 headline only (steps 1 and 3 of profile_round.sh): bash tools/profile_eval.sh r04
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_${TAG}e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PY=python3
export EBO_LIB_PATH=$R/event-based-odomety_amd/libebo_hip.so
S="$PY $R/event-based-odomety_amd/tools/pmc_summary.py"
rocprofv3 --kernel-trace --stats -d $O/ks_noextras --output-format csv -- $PY $R/bench.py --no-extras --steps 100 --warmup 5 > $O/bench_noextras.json 2> $O/bench_noextras.err
cp $O/ks_noextras/*/*kernel_stats.csv $O/${TAG}_bench_noextras_kernel_stats.csv 2>/dev/null
B="$PY $R/bench.py --no-extras --steps 5 --warmup 1 --preheat 0 --cpu-seconds 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/e3a --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT -d $O/e3b --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/e3c --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/e3d --output-format csv -- $B > /dev/null 2>&1
$S k_eval3 $O/e3a $O/e3b $O/e3c $O/e3d > $O/${TAG}_pmc_k_eval3.txt
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
grep -E "FETCH|WRITE|ms under" $O/${TAG}_pmc_k_eval3.txt | head; head -2 $O/${TAG}_bench_noextras_kernel_stats.csv | cut -c1-200
