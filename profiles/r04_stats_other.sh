#!/bin/bash
# rocprofv3 kernel stats of the two constant-heavy 10 GiB configurations (final engine)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in csv2json iso_datetime_to_json; do
  rm -rf /tmp/kt_$P && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$P -o p -- python $R/bench.py --program $P --steps 3 --warmup 1 --no-cpu > $O/bench_under_rocprof_$P.log 2>&1
  cp $(find /tmp/kt_$P -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$P.csv
  grep "^{\"metric\"" $O/bench_under_rocprof_$P.log | tail -1 > $O/bench_under_rocprof_$P.json
  head -6 $O/kernel_stats_$P.csv | cut -c1-160
done
