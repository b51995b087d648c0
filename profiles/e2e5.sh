#!/bin/bash
# kx_run_fd: is the slow mode of back-to-back runs the previous process's teardown (VRAM being cleared)?  Pause between runs.
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/e2e5; mkdir -p $O; cd $R
python - <<'PY'
import sys; sys.path.insert(0, ".")
from kleenexlang_amd import workloads
base = workloads.generate("apache_log", 32 << 20, 0x4B4C4558)
with open("/tmp/log16g", "wb") as f:
    for _ in range(512): f.write(base)
PY
kleenexlang_amd/_build/kexc compile --quiet kleenexlang_amd/programs/apache_log.kex --out /tmp/apache_bin
cat /tmp/log16g > /dev/null
: > $O/e2e.txt
for envs in "A=1" "A=1" "A=1" "KX_READ_THREADS=8" "KX_READ_THREADS=8" "KX_READ_THREADS=16" "KX_READ_THREADS=16" "KX_READ_THREADS=8 KX_WINDOW_BYTES=536870912" "KX_READ_THREADS=8 KX_WINDOW_BYTES=268435456"; do
  sleep 5
  echo "after 5 s pause; env: $envs" >> $O/e2e.txt
  env $envs KX_FD_TRACE=1 /tmp/apache_bin -t < /tmp/log16g > /dev/null 2>> $O/e2e.txt
done
echo "back to back:" >> $O/e2e.txt
env KX_FD_TRACE=1 /tmp/apache_bin -t < /tmp/log16g > /dev/null 2>> $O/e2e.txt
cat $O/e2e.txt
