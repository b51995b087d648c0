#!/bin/bash
# Round 6, second A/B on one box: merged constants (window J = KX_DF_J) x cooperative record flush in k_dforward (KX_DEBUG_FLAGS=256: off);
# every output byte checked by bench.py
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab2.txt
: > $OUT
run() {  # J flags program
  KX_DF_J=$1 KX_DEBUG_FLAGS=$2 python bench.py --program $3 --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('J=$1 flags=$2', '$3', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
}
for rep in 1 2; do
  run 0 256 apache_log
  run 0 0 apache_log
  run 1 0 apache_log
  run 2 0 apache_log
  run 2 256 apache_log
  run 6 0 apache_log
done
for p in csv2json iso_datetime_to_json thousand_sep; do
  run 0 256 $p
  run 0 0 $p
  run 2 0 $p
done
cat $OUT
