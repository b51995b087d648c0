#!/bin/bash
# soak of the job-stride layout: the random-program soak with KX_JL=1 (every qualifying program on that layout) and the windowed soak
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
KX_JL=1 SOAK_LO=7000 SOAK_HI=8200 timeout 1500 python tests/soak/soak_engine.py > $O/soak_engine_jl1.txt 2>&1; tail -2 $O/soak_engine_jl1.txt
KX_JL=1 timeout 1200 python tests/soak/soak_windows.py > $O/soak_windows_jl1.txt 2>&1; tail -1 $O/soak_windows_jl1.txt
KX_JL=1 timeout 900 python tests/soak/soak_coders.py > $O/soak_coders_jl1.txt 2>&1; tail -1 $O/soak_coders_jl1.txt
