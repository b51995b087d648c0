#!/bin/bash
# Round 6, fourth A/B on one box: non-temporal loads of the forward pass's input lines and non-temporal stores of k_demit's flush
# (_probe/r06/nt/libkxhip.so: the tree's engine with those two lines changed) against the tree's engine, alternating
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab4.txt
: > $OUT
cp kleenexlang_amd/_build/libkxhip.so /tmp/tree.so
run() {  # which program
  if [ $1 = nt ]; then cp _probe/r06/nt/libkxhip.so kleenexlang_amd/_build/libkxhip.so; else cp /tmp/tree.so kleenexlang_amd/_build/libkxhip.so; fi
  python bench.py --program $2 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
}
for rep in 1 2 3; do run tree apache_log; run nt apache_log; done
for p in csv2json iso_datetime_to_json; do run tree $p; run nt $p; done
cp /tmp/tree.so kleenexlang_amd/_build/libkxhip.so
cat $OUT
