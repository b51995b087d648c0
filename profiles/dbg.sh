#!/bin/bash
# small, safe debug run: no core dumps, tiny tests only
ulimit -c 0
export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dbg; mkdir -p $O; cd $R
for f in 0; do
  echo "== KX_DEBUG_FLAGS=$f" >> $O/log.txt
  KX_DEBUG_FLAGS=$f timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "register_actions" 2>&1 | tail -15 >> $O/log.txt
done
df -h . | tail -1 >> $O/log.txt
tail -40 $O/log.txt
