#!/bin/bash
# round 2, first GPU call: LDS cost table, RCCL probe, GPU test suite, bench lines of the three 10 GiB configs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 300 ./_probe/lds_cost > $O/lds_cost.txt 2>&1
echo "== rccl world 2 single device" > $O/rccl.txt
timeout 600 python bench.py --gpus 2 --single-device --gib 1 --steps 3 --warmup 1 --no-cpu >> $O/rccl.txt 2>&1
echo "rc=$?" >> $O/rccl.txt
echo "== rccl world 4 single device" >> $O/rccl.txt
timeout 600 python bench.py --gpus 4 --single-device --gib 1 --steps 3 --warmup 1 --no-cpu >> $O/rccl.txt 2>&1
echo "rc=$?" >> $O/rccl.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "rc=$?" >> $O/pytest.txt
for p in apache_log csv2json iso_datetime_to_json; do
  timeout 600 python bench.py --program $p --steps 10 --warmup 2 > $O/bench_$p.json 2> $O/bench_$p.err
done
tail -3 $O/pytest.txt; tail -5 $O/rccl.txt; cat $O/bench_apache_log.json | cut -c1-600
