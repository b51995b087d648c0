#!/bin/bash
# end-to-end `BIN -t < file > /dev/null` of the produced binary (windows pipelined: read || compute || write)
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-e2e}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "binary or windows or phase or multi_stage" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
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
for w in "" 268435456 4294967296; do
  for i in 1 2; do
    echo "window=${w:-default(1GiB)} run $i" >> $O/e2e.txt
    KX_WINDOW_BYTES=$w /tmp/apache_bin -t < /tmp/log8g > /dev/null 2>> $O/e2e.txt
  done
done
echo "pipe:" >> $O/e2e.txt; cat /tmp/log8g | /tmp/apache_bin -t > /dev/null 2>> $O/e2e.txt
echo "md5 of output (default windows vs one window):" >> $O/e2e.txt
head -c 3221225472 /tmp/log8g > /tmp/log3g
/tmp/apache_bin < /tmp/log3g | md5sum >> $O/e2e.txt
KX_WINDOW_BYTES=17179869184 /tmp/apache_bin < /tmp/log3g | md5sum >> $O/e2e.txt
cat $O/e2e.txt
