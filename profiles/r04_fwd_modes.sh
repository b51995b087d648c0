#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r04o; mkdir -p $O
for pad in 0 0 0 0 4096 65536 1048576 3145728 0 0; do timeout 300 python profiles/r04_fwd_modes.py $pad 2>/dev/null | tee -a $O/modes.txt; done
