#!/bin/bash
# Round 6, first A/B on one box: the round-start walk (_probe/r06/base: gen_sweeps.py with KX_GEN_JUNK=0) against junk-tolerant byte
# stores, and merged constants with windows J = 0 / 2 / 6 (KX_DF_J), alternating; every output byte checked by bench.py
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab1.txt
: > $OUT
cp kleenexlang_amd/_build/libkxhip.so /tmp/new.so
run() {  # which J program
  if [ $1 = base ]; then cp _probe/r06/base/libkxhip.so kleenexlang_amd/_build/libkxhip.so; else cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so; fi
  KX_DF_J=$2 python bench.py --program $3 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 J=$2', '$3', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
}
for rep in 1 2; do
  run base 0 apache_log
  run new 0 apache_log
  run new 2 apache_log
  run new 6 apache_log
done
for p in csv2json iso_datetime_to_json; do
  run base 0 $p
  run new 0 $p
  run new 2 $p
  run new 6 $p
done
cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so
cat $OUT
