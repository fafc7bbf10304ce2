# A/B of two builds on the count-image kernels (>= 1 GiB working sets): libebo_hip_prof.so (variant) against libebo_hip.so
set -e
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
echo "# prof = $1; hip = the shipped build"
for rep in 1 2; do
for lib in libebo_hip_prof.so libebo_hip.so; do
  echo "== $lib"
  for cfg in "2 1536" "3 512" "4 72"; do
    EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 300 python $T/ab_count.py $cfg "" 2>&1 | grep -v amdgpu.ids
  done
done
done
