cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab6.txt
: > $OUT
run() {  # label program env...
  label=$1; prog=$2; shift 2
  env "$@" python bench.py --program $prog --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', '$prog', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
}
for p in apache_log csv2json iso_datetime_to_json; do
  run W12 $p X=1
  run W16 $p KX_EMIT_WAVES=16
  run W8 $p KX_EMIT_WAVES=8
done
run seg32k apache_log X=1
python bench.py --program apache_log --segment 32768 --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg32768', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
python bench.py --program apache_log --segment 20480 --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg20480', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])" >> $OUT
cat $OUT
