for p in apache_log csv2json iso_datetime_to_json; do for seg in 0 24576 32768 49152 65536; do
  echo -n "$p seg=$seg: "; timeout 300 python bench.py --program $p --segment $seg --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])"
done; done
