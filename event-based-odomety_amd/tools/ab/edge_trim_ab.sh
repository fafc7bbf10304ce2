#!/bin/bash
# I stored for the rows with taps only (round 5, late): bits against the build before it, then timings of the two builds,
# twice.  libebo_hip_base.so = the sources of the commit to compare with, built beside the tree's own libraries:
#   git stash / git worktree of that commit; make -C event-based-odomety_amd/csrc ../libebo_hip_base.so OUT=../libebo_hip_base.so \
#        OBJDIR=build_base FLAGS="-std=c++17 -O3 -fPIC -ffp-contract=off -DEBO_AB"   (not kept in the tree)
set -e
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
P=$GRAFT_REPO_ROOT/event-based-odomety_amd
mkdir -p gpurun_out/trim
EBO_LIB_PATH=$P/libebo_hip_base.so timeout -k 10 300 python $T/ab/edge_bits_dump.py gpurun_out/trim/base.npz 2>&1 | grep -v amdgpu.ids
EBO_LIB_PATH=$P/libebo_hip_ab.so timeout -k 10 300 python $T/ab/edge_bits_dump.py gpurun_out/trim/new.npz 2>&1 | grep -v amdgpu.ids
python $T/ab/edge_bits_cmp.py gpurun_out/trim/base.npz gpurun_out/trim/new.npz
for rep in 1 2; do
for lib in libebo_hip_base.so libebo_hip_ab.so; do
  echo "== $lib"
  for cfg in "0 256" "3 64" "2 64" "4 8"; do
    EBO_LIB_PATH=$P/$lib timeout -k 10 300 python $T/ab_edge.py $cfg "" 2>&1 | grep -v amdgpu.ids
  done
  EBO_AB_FLOWSCALE=1.0 EBO_LIB_PATH=$P/$lib timeout -k 10 300 python $T/ab_edge.py 0 256 "" 2>&1 | grep -v amdgpu.ids
  EBO_AB_FLOWSCALE=0.0 EBO_LIB_PATH=$P/$lib timeout -k 10 300 python $T/ab_edge.py 0 256 "" 2>&1 | grep -v amdgpu.ids
done
done
