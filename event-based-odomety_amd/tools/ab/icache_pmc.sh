#!/bin/bash
# Instruction-cache / instruction-fetch counters of k_eval_edge on the reference-default configuration x 256 windows
# (one --pmc pass per counter group; nothing else traced).  usage (through gpurun, from the repo root):
#   bash event-based-odomety_amd/tools/ab/icache_pmc.sh <tag>
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/icache_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EBO_LIB_PATH=$R/event-based-odomety_amd/libebo_hip.so
PY=python3
S="$PY $R/event-based-odomety_amd/tools/pmc_summary.py"
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -o -i "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*" $O/list_avail.txt | sort -u > $O/names.txt
E="$PY $R/event-based-odomety_amd/tools/ab_edge.py 0 256"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_REQ" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/p$i --output-format csv -- $E > $O/p$i.log 2>&1
  echo "group $i rc=$?: $grp"
  $S k_eval_edge $O/p$i >> $O/${TAG}_icache_k_eval_edge.txt 2>&1
done
find $O -name "*counter_collection.csv" -size +2M -delete
cat $O/${TAG}_icache_k_eval_edge.txt
