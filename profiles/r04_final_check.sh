#!/bin/bash
# after the compiler's regex-range change (csv2json: 27 states): the whole GPU suite again and the csv2json bench line
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04zz; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_pytest.txt 2>&1; tail -2 $O/gpu_pytest.txt
timeout 900 python bench.py --program csv2json --steps 10 --warmup 2 --no-cpu > $O/bench_csv2json.json 2> $O/bench_csv2json.err; cut -c1-400 $O/bench_csv2json.json
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
