#!/bin/bash
# kx_run_fd: where the wall time of `BIN -t < file > /dev/null` goes (KX_FD_TRACE=1), 17 GB apache_log from the page cache
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/e2e3; mkdir -p $O; cd $R
python - <<'PY'
import sys; sys.path.insert(0, ".")
from kleenexlang_amd import workloads
base = workloads.generate("apache_log", 32 << 20, 0x4B4C4558)
with open("/tmp/log16g", "wb") as f:
    for _ in range(512): f.write(base)
print("file bytes", 512 * len(base))
PY
kleenexlang_amd/_build/kexc compile --quiet kleenexlang_amd/programs/apache_log.kex --out /tmp/apache_bin
cat /tmp/log16g > /dev/null
numactl --hardware > $O/numa.txt 2>&1; lscpu | head -20 >> $O/numa.txt
rocm-smi --showtopo >> $O/numa.txt 2>&1
: > $O/e2e.txt
for cfg in "4 1073741824" "4 1073741824" "8 1073741824" "16 268435456" "$@"; do
  set -- $cfg
  echo "threads=$1 window=$2 $3" >> $O/e2e.txt
  env $3 KX_FD_TRACE=1 KX_READ_THREADS=$1 KX_WINDOW_BYTES=$2 /tmp/apache_bin -t < /tmp/log16g > /dev/null 2>> $O/e2e.txt
done
cat $O/e2e.txt
