#!/bin/bash
# round-2 evidence run: full GPU test suite, bench lines of the 10 GiB configs, rocprofv3 stats + PMC traffic (collect.sh)
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r02z}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
for p in apache_log csv2json iso_datetime_to_json; do
  timeout 900 python bench.py --program $p --steps 20 --warmup 5 > $O/bench_$p.json 2> $O/bench_$p.err
  python -c "
import json; d=json.loads(open('$O/bench_$p.json').read()); print('$p', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['output_checked_bit_exact'], d.get('cpu_baseline'))" 2>&1 | cut -c1-400
done
bash profiles/collect.sh ${TAG}_prof > $O/collect.log 2>&1; tail -3 $O/collect.log
KX_SPARSE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu > $O/bench_sparse.json 2> $O/bench_sparse.err
python -c "
import json; d=json.loads(open('$O/bench_sparse.json').read()); print('sparse', d['value'], d['kernels_ms'], d['output_checked_bit_exact'])"
