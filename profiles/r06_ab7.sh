#!/bin/bash
# Round 6, A/B 7: heads and tails of constants as b8 + b16 stores (8 store levels per 16 bytes instead of 10) against the engine before (_probe/r06/prev)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab7.txt
: > $OUT
cp kleenexlang_amd/_build/libkxhip.so /tmp/tree.so
run() {
  if [ $1 = prev ]; then cp _probe/r06/prev/libkxhip.so kleenexlang_amd/_build/libkxhip.so; else cp /tmp/tree.so kleenexlang_amd/_build/libkxhip.so; fi
  python bench.py --program $2 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
}
for rep in 1 2; do for p in apache_log csv2json iso_datetime_to_json; do run prev $p; run new $p; done; done
cp /tmp/tree.so kleenexlang_amd/_build/libkxhip.so
cat $OUT
