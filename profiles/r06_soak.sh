#!/bin/bash
# Round 6 soak on the GPU box: seeded random programs on the engine vs the oracle, the delayed form in its three settings
# (default; KX_DF=2 = taken whatever the share of undecided contexts, so that fall-backs in mid-run are exercised; delay 1), the
# general engine alone, windowed binaries.  Round 6: KX_DF=2 now also exercises the exact slow path (an undecided context arms it) on random programs.   usage: profiles/r06_soak.sh TAG
TAG=${1:-r06_soak}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
SOAK_LO=12000 SOAK_HI=12500 timeout 700 python tests/soak/soak_engine.py > $OUT/soak_engine_default.txt 2>&1
KX_DF=2 SOAK_LO=12500 SOAK_HI=13000 timeout 700 python tests/soak/soak_engine.py > $OUT/soak_engine_df2.txt 2>&1
KX_DF=2 KX_DF_K=1 SOAK_LO=13000 SOAK_HI=13400 timeout 700 python tests/soak/soak_engine.py > $OUT/soak_engine_df2_k1.txt 2>&1
KX_DF=0 SOAK_LO=13400 SOAK_HI=13700 timeout 700 python tests/soak/soak_engine.py > $OUT/soak_engine_general.txt 2>&1
timeout 700 python tests/soak/soak_windows.py > $OUT/soak_windows_default.txt 2>&1
KX_DF=2 timeout 700 python tests/soak/soak_windows.py > $OUT/soak_windows_df2.txt 2>&1
tail -n 2 $OUT/*.txt
