#!/bin/bash
# kernel trace of the single-window reference-default call (edge + TV, global LM): which launches make its 0.75 ms
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sw
mkdir -p $OUT
python3 $R/event-based-odomety_amd/tools/time_reference_call.py 0 1 > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof -o sw --output-format csv -- python3 $R/event-based-odomety_amd/tools/time_reference_call.py 0 1 > $OUT/prof.txt 2>&1
ls -R $OUT/prof | head -30
