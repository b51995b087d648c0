#!/bin/bash
# kx_run_fd through pipes (1 MiB pipe buffers): cat file | BIN -t > /dev/null, BIN -t < file | cat > /dev/null, both
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/e2e6; mkdir -p $O; cd $R
python - <<'PY'
import sys; sys.path.insert(0, ".")
from kleenexlang_amd import workloads
base = workloads.generate("apache_log", 32 << 20, 0x4B4C4558)
with open("/tmp/log8g", "wb") as f:
    for _ in range(256): f.write(base)
print("file bytes", 256 * len(base))
PY
kleenexlang_amd/_build/kexc compile --quiet kleenexlang_amd/programs/apache_log.kex --out /tmp/apache_bin
cat /tmp/log8g > /dev/null
: > $O/e2e.txt
for i in 1 2; do
  sleep 5; echo "pipe in:  cat file | BIN -t > /dev/null" >> $O/e2e.txt
  cat /tmp/log8g | KX_FD_TRACE=1 /tmp/apache_bin -t > /dev/null 2>> $O/e2e.txt
  sleep 5; echo "pipe out: BIN -t < file | cat > /dev/null" >> $O/e2e.txt
  KX_FD_TRACE=1 /tmp/apache_bin -t < /tmp/log8g 2>> $O/e2e.txt | cat > /dev/null
  sleep 5; echo "both:     cat file | BIN -t | cat > /dev/null" >> $O/e2e.txt
  cat /tmp/log8g | KX_FD_TRACE=1 /tmp/apache_bin -t 2>> $O/e2e.txt | cat > /dev/null
done
sleep 5; echo "file to file on /tmp:" >> $O/e2e.txt
KX_FD_TRACE=1 /tmp/apache_bin -t < /tmp/log8g > /tmp/out8g 2>> $O/e2e.txt
ls -l /tmp/out8g | awk '{print $5}' >> $O/e2e.txt
cat $O/e2e.txt
