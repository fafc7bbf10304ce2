#!/bin/bash
# a launch smaller than the chip (1 / 2 windows of the reference default = 108 / 216 units on 256 CUs): lanes per unit
set -e
cd ${GRAFT_REPO_ROOT:-.}
S='"" "EBO_EDGE_BLOCK=512" "EBO_EDGE_BLOCK=768" "EBO_EDGE_BLOCK=512,EBO_EDGE_LDS_KB=160" "EBO_EDGE_BLOCK=768,EBO_EDGE_LDS_KB=160" "EBO_EDGE_BLOCK=128" ""'
for w in 1 2 4; do
  eval python event-based-odomety_amd/tools/ab_edge.py 0 $w $S
done
