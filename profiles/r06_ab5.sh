cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab5.txt
: > $OUT
cp kleenexlang_amd/_build/libkxhip.so /tmp/tree.so
run() {
  if [ $1 = nt ]; then cp _probe/r06/nt/libkxhip.so kleenexlang_amd/_build/libkxhip.so; else cp /tmp/tree.so kleenexlang_amd/_build/libkxhip.so; fi
  python bench.py --program apache_log --escapes $2 --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'esc$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
}
for rep in 1 2; do run tree 100; run nt 100; run tree 1000; run nt 1000; done
cp /tmp/tree.so kleenexlang_amd/_build/libkxhip.so
cat $OUT
