#!/bin/bash
# Dynamic instruction mix of k_eval_edge PER PHASE on the reference-default configuration x 256 windows: the A/B build's
# ablation chain (EBO_EDGE_ABLATE: 8 = no gather, 12 = no gather + no reverse, 14 = + no window maxima, 15 = + no
# eigenvalue pass) under --pmc passes of the instruction-class counters; a phase's mix is the difference of two
# neighbours of the chain (tools/ab/edge_phase_mix.py prints the table).  One --pmc pass per group, nothing traced.
#   bash event-based-odomety_amd/tools/ab/edge_phase_mix.sh <tag>      (through gpurun, from the repo root)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/edge_mix_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EBO_LIB_PATH=$R/event-based-odomety_amd/libebo_hip_ab.so
PY=python3
S="$PY $R/event-based-odomety_amd/tools/pmc_summary.py"
G1="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
G2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_BRANCH SQ_INSTS_VMEM"
G3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
for ab in 0 8 12 14 15; do
  E="$PY $R/event-based-odomety_amd/tools/ab_edge.py 0 256 EBO_EDGE_ABLATE=$ab"
  i=0
  for grp in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    rocprofv3 --pmc $grp -d $O/a${ab}_g$i --output-format csv -- $E > $O/a${ab}_g$i.log 2>&1
    echo "ablate $ab group $i rc=$?"
    $S k_eval_edge $O/a${ab}_g$i >> $O/a${ab}.txt 2>&1
  done
done
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
$PY $R/event-based-odomety_amd/tools/ab/edge_phase_mix.py $O > $O/${TAG}_edge_phase_mix.txt
cat $O/${TAG}_edge_phase_mix.txt
