#!/bin/bash
# k_edge_classify (the bounding-box pass and the second launch's list in one small kernel up front) on / off, A/B build;
# bits of the new build against libebo_hip_base.so first (see tools/ab/edge_trim_ab.sh for how that library is built)
set -e
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
P=$GRAFT_REPO_ROOT/event-based-odomety_amd
mkdir -p gpurun_out/trim
EBO_LIB_PATH=$P/libebo_hip_base.so timeout -k 10 300 python $T/ab/edge_bits_dump.py gpurun_out/trim/base.npz 2>&1 | grep -v amdgpu.ids
EBO_LIB_PATH=$P/libebo_hip_ab.so timeout -k 10 300 python $T/ab/edge_bits_dump.py gpurun_out/trim/new.npz 2>&1 | grep -v amdgpu.ids
EBO_EDGE_CLASSIFY=0 EBO_LIB_PATH=$P/libebo_hip_ab.so timeout -k 10 300 python $T/ab/edge_bits_dump.py gpurun_out/trim/new0.npz 2>&1 | grep -v amdgpu.ids
python $T/ab/edge_bits_cmp.py gpurun_out/trim/base.npz gpurun_out/trim/new.npz | grep -v identical
python $T/ab/edge_bits_cmp.py gpurun_out/trim/new0.npz gpurun_out/trim/new.npz | grep -v identical
export EBO_LIB_PATH=$P/libebo_hip_ab.so
S='"EBO_EDGE_CLASSIFY=0" "EBO_EDGE_CLASSIFY=1" "EBO_EDGE_CLASSIFY=0" "EBO_EDGE_CLASSIFY=1"'
eval python $T/ab_edge.py 0 256 $S 2>&1 | grep -v amdgpu.ids
eval python $T/ab_edge.py 3 64 $S 2>&1 | grep -v amdgpu.ids
eval python $T/ab_edge.py 0 32 $S 2>&1 | grep -v amdgpu.ids
EBO_AB_FLOWSCALE=1.0 eval python $T/ab_edge.py 0 256 $S 2>&1 | grep -v amdgpu.ids
EBO_AB_FLOWSCALE=0.0 eval python $T/ab_edge.py 0 256 $S 2>&1 | grep -v amdgpu.ids
for lib in libebo_hip_base.so libebo_hip_ab.so; do
  echo "== $lib"
  for cfg in "0 256" "3 64" "2 64" "4 8" "0 1"; do
    EBO_LIB_PATH=$P/$lib timeout -k 10 300 python $T/ab_edge.py $cfg "" 2>&1 | grep -v amdgpu.ids
  done
done
