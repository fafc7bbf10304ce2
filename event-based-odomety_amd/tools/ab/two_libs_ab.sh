# A/B of two builds of the library on one GPU box: libebo_hip_prof.so (the variant, built by hand with OUT=/OBJDIR=/FLAGS=)
# against libebo_hip.so (the shipped build), same process order twice.
#   bash event-based-odomety_amd/tools/ab/two_libs_ab.sh "<what prof is>"
set -e
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
echo "# prof = $1; hip = the shipped build"
for rep in 1 2; do
for lib in libebo_hip_prof.so libebo_hip.so; do
  echo "== $lib"
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 3 64 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_edge.py 0 256 "" 2>&1 | grep -v amdgpu.ids
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_edge.py 3 64 "" 2>&1 | grep -v amdgpu.ids
done
done
for lib in libebo_hip_prof.so libebo_hip.so; do
  echo "== $lib"
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 2 256 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 4 8 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 0 256 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 300 python $T/time_reference_call.py 2>&1 | grep -v amdgpu.ids | tail -4
done
