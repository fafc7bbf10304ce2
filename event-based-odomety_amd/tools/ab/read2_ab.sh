# A/B of paired (ds_read2_b64) against single (ds_read_b64) LDS reads: libebo_hip_prof.so = the default build with
# -DEBO_LDS_PAIRED_READS (csrc/ebo_kernels.hip: lds_ld), libebo_hip.so = the shipped build.  Run on the GPU box:
#   make -C event-based-odomety_amd/csrc OUT=../libebo_hip_prof.so OBJDIR=build_prof FLAGS="<FLAGS> -DEBO_LDS_PAIRED_READS" ../libebo_hip_prof.so
#   bash event-based-odomety_amd/tools/ab/read2_ab.sh
set -e
cd $GRAFT_REPO_ROOT
T=event-based-odomety_amd/tools
for rep in 1 2; do
for lib in libebo_hip_prof.so libebo_hip.so; do
  echo "== $lib (prof = paired ds_read2_b64 reads, hip = single ds_read_b64)"
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 3 64 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_edge.py 0 256 "" 2>&1 | grep -v amdgpu.ids
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_edge.py 3 64 "" 2>&1 | grep -v amdgpu.ids
done
done
for lib in libebo_hip_prof.so libebo_hip.so; do
  echo "== $lib"
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 2 256 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_eval.py 4 8 "" 2>&1 | grep cfg
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 200 python $T/ab_edge.py 2 64 "" 2>&1 | grep -v amdgpu.ids
  EBO_LIB_PATH=$GRAFT_REPO_ROOT/event-based-odomety_amd/$lib timeout -k 10 300 python $T/time_reference_call.py 2>&1 | grep -v amdgpu.ids | tail -8
done
