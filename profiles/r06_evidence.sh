#!/bin/bash
# Round-6 evidence of the tree as it stands (GPU box): the GPU suite, the bench lines of the 10 GiB configurations (and of the log with
# escaped quotes), rocprofv3 kernel stats + FETCH/WRITE_SIZE of the default bench (collect.sh), SQ counters of the delayed-form kernels
# (collect_sq_df.sh), L2-side counters of the forward pass with both record flushes (collect_tcc.sh), the phase timeline of k_demit.
# usage: profiles/r06_evidence.sh TAG
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/gpu_pytest.txt
for P in apache_log csv2json iso_datetime_to_json thousand_sep; do
  timeout 300 python bench.py --steps 10 --warmup 2 --program $P $( [ $P = apache_log ] || echo --no-cpu ) > $OUT/bench_$P.json 2> $OUT/bench_$P.err
done
KX_DF=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_apache_log_general_engine.json 2>> $OUT/bench_apache_log.err
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --escapes 100 > $OUT/bench_apache_log_escapes100.json 2>> $OUT/bench_apache_log.err
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --escapes 1000 > $OUT/bench_apache_log_escapes1000.json 2>> $OUT/bench_apache_log.err
KX_NO_SLOW=1 KX_DF_BACKOFF_OFF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --escapes 100 > $OUT/bench_apache_log_escapes100_round5_fallback.json 2>> $OUT/bench_apache_log.err
KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | grep "timeline" | tail -1 > $OUT/timeline.txt
profiles/collect.sh $TAG > $OUT/collect.log 2>&1
profiles/collect_sq_df.sh $TAG > $OUT/sq.log 2>&1
profiles/collect_tcc.sh $TAG > $OUT/tcc.log 2>&1
tail -3 $OUT/gpu_pytest.txt; tail -3 $OUT/sq.log; tail -2 $OUT/tcc.log; cat $OUT/timeline.txt
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["ms_per_step_median"], d["kernels_ms"], d["roofline"]["frac"], d["output_checked_bit_exact"], d.get("escaped_quotes_injected"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
