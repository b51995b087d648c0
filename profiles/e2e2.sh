#!/bin/bash
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/e2e2; mkdir -p $O; cd $R
python - <<'PY'
import sys; sys.path.insert(0, ".")
from kleenexlang_amd import workloads
base = workloads.generate("apache_log", 32 << 20, 0x4B4C4558)
with open("/tmp/log16g", "wb") as f:
    for _ in range(512): f.write(base)
print("file bytes", 512 * len(base))
PY
free -g | head -2 >> $O/e2e.txt; nproc >> $O/e2e.txt
kleenexlang_amd/_build/kexc compile --quiet kleenexlang_amd/programs/apache_log.kex --out /tmp/apache_bin
cat /tmp/log16g > /dev/null
for cfg in "4 1073741824" "8 1073741824" "8 2147483648" "16 1073741824" "2 1073741824" "8 536870912"; do
  set -- $cfg
  for i in 1 2; do
    echo "threads=$1 window=$2 run $i" >> $O/e2e.txt
    KX_READ_THREADS=$1 KX_WINDOW_BYTES=$2 /tmp/apache_bin -t < /tmp/log16g > /dev/null 2>> $O/e2e.txt
  done
done
( time cat /tmp/log16g > /dev/null ) 2>> $O/e2e.txt
cat $O/e2e.txt
