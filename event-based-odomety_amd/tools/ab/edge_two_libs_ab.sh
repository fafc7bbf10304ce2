# A/B of two builds on the edge-loss kernels: libebo_hip_prof.so (variant) against libebo_hip.so (shipped), twice
set -e
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
echo "# prof = $1; hip = the shipped build"
for rep in 1 2; do
for lib in libebo_hip_prof.so libebo_hip.so; do
  echo "== $lib"
  for cfg in "0 256" "2 64" "3 64" "4 8"; do
    EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 300 python $T/ab_edge.py $cfg "" 2>&1 | grep -v amdgpu.ids
  done
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 300 python $T/time_reference_call.py 2>&1 | grep -v amdgpu.ids | tail -4
done
done
