#!/bin/bash
# Collects the rocprofv3 evidence of a round on one MI355X (run through gpurun from the repo root):
#   bash event-based-odomety_amd/tools/profile_round.sh r05
# Writes summaries under gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/.
# Counter passes are separate runs (--pmc never together with traces), the program itself follows `--`.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PY=python3
# the SHIPPED library for every tool (tools/ab_*.py default to the -DEBO_AB build, whose kernels carry the diagnostics)
export EBO_LIB_PATH=$R/event-based-odomety_amd/libebo_hip.so
S="$PY $R/event-based-odomety_amd/tools/pmc_summary.py"

# 1. kernel stats of the timed region of the bench (the dominant kernel's average launch duration)
rocprofv3 --kernel-trace --stats -d $O/ks_noextras --output-format csv -- $PY $R/bench.py --no-extras --steps 100 --warmup 5 > $O/bench_noextras.json 2> $O/bench_noextras.err
cp $O/ks_noextras/*/*kernel_stats.csv $O/${TAG}_bench_noextras_kernel_stats.csv 2>/dev/null
echo "[1/6] bench --no-extras done"

# 2. every kernel of the path: the default bench with its extras
rocprofv3 --kernel-trace --stats -d $O/ks_full --output-format csv -- $PY $R/bench.py --steps 20 --warmup 2 --cpu-seconds 2 > $O/bench_full.json 2> $O/bench_full.err
cp $O/ks_full/*/*kernel_stats.csv $O/${TAG}_bench_full_kernel_stats.csv 2>/dev/null
echo "[2/6] bench full done"

# 3. counters of the dominant kernel at the shipping configuration (C3 x 64 windows)
B="$PY $R/bench.py --no-extras --steps 5 --warmup 1 --preheat 0 --cpu-seconds 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/e3a --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT -d $O/e3b --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/e3c --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/e3d --output-format csv -- $B > /dev/null 2>&1
$S k_eval3 $O/e3a $O/e3b $O/e3c $O/e3d > $O/${TAG}_pmc_k_eval3.txt
echo "[3/6] k_eval3 counters done"

# 4. the edge kernel on the reference-default configuration x 256 windows
E="$PY $R/event-based-odomety_amd/tools/ab_edge.py 0 256"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/ea --output-format csv -- $E > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT -d $O/eb --output-format csv -- $E > /dev/null 2>&1
$S k_eval_edge $O/ea $O/eb > $O/${TAG}_pmc_k_eval_edge.txt
rocprofv3 --kernel-trace --stats -d $O/ks_edge --output-format csv -- $E > $O/ab_edge.log 2>&1
cp $O/ks_edge/*/*kernel_stats.csv $O/${TAG}_edge_kernel_stats.csv 2>/dev/null
echo "[4/6] k_eval_edge counters done"

# 5. the count-image kernels on >= 1 GiB working sets (C2 x 1536, C3 x 512, C4 x 72 windows)
for cfg in "2 1536" "3 512" "4 72"; do
  n=$(echo $cfg | tr ' ' 'x')
  C="$PY $R/event-based-odomety_amd/tools/ab_count.py $cfg"
  rocprofv3 --kernel-trace --stats -d $O/ks_count_$n --output-format csv -- $C > $O/ab_count_$n.log 2>&1
  cp $O/ks_count_$n/*/*kernel_stats.csv $O/${TAG}_count_${n}_kernel_stats.csv 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE -d $O/cf_$n --output-format csv -- $C > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/cw_$n --output-format csv -- $C > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/cs_$n --output-format csv -- $C > /dev/null 2>&1
  $S k_count $O/cf_$n $O/cw_$n $O/cs_$n > $O/${TAG}_pmc_count_$n.txt
done
echo "[5/6] count kernels done"

# 5b. the N > 1 default workload's kernel (k_solve_independent, config-4 shard of one GPU) at N = 1: kernel stats + HBM traffic
C4="$PY $R/bench.py --workload c4 --steps 5 --warmup 1 --preheat 0 --cpu-seconds 1"
rocprofv3 --kernel-trace --stats -d $O/ks_c4 --output-format csv -- $C4 > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
cp $O/ks_c4/*/*kernel_stats.csv $O/${TAG}_bench_c4_1gpu_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $O/c4f --output-format csv -- $C4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/c4w --output-format csv -- $C4 > /dev/null 2>&1
$S k_solve_independent $O/c4f $O/c4w > $O/${TAG}_pmc_k_solve_independent_c4.txt
echo "[5b] c4 solve kernel done"

# 6. the N > 1 code paths rehearsed on this one GPU (2 ranks share it, collectives over gloo)
cd $R
EBO_BENCH_REHEARSE=1 $PY bench.py --gpus 2 --steps 5 --warmup 1 > $O/${TAG}_bench_gpus2_c4_rehearsal.json 2> $O/bench_gpus2.err
EBO_BENCH_REHEARSE=1 $PY bench.py --gpus 2 --replicas --steps 5 --warmup 1 > $O/${TAG}_bench_gpus2_replicas_rehearsal.json 2> $O/bench_gpus2r.err
echo "[6/6] rehearsals done"
# keep the merge small: raw counter CSVs are large
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
