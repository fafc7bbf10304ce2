#!/bin/bash
# from how many units on does the compact path (two launches, three workgroups per CU) pay?  lock-step reference-default
# call and batched evaluation at 4 .. 32 windows, compact forced / never / the shipped rule (A/B build)
cd $GRAFT_REPO_ROOT
export EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/libebo_hip_ab.so
T=event-based-odomety_amd/tools
for w in 4 8 12 16 24 32; do
  python $T/ab_edge.py 0 $w "EBO_EDGE_COMPACT=0" "EBO_EDGE_COMPACT=1" "EBO_EDGE_COMPACT=0" "EBO_EDGE_COMPACT=1" 2>&1 | grep -v amdgpu.ids
done
for c in 0 1; do
  echo "== lock-step call, EBO_EDGE_COMPACT=$c"
  EBO_EDGE_COMPACT=$c python $T/time_reference_call.py 0 4 8 16 32 64 2>&1 | grep -v amdgpu.ids
done
echo "== lock-step call, shipped rule"
python $T/time_reference_call.py 0 4 8 16 32 64 2>&1 | grep -v amdgpu.ids
