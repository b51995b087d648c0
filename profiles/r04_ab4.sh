#!/bin/bash
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04f}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
KX_JL=1 timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_pipeline_seam.py tests/test_regex_coder.py -m gpu -x -q -k "not 10gib and not rccl" > $O/pytest_jl1.txt 2>&1; echo "rc=$?" >> $O/pytest_jl1.txt; tail -3 $O/pytest_jl1.txt
run() { # name env...
  local name=$1; shift
  env "$@" KX_DEBUG=1 timeout 600 python bench.py --program $P --steps 5 --warmup 1 --no-cpu > $O/bench_${P}_$name.json 2> $O/bench_${P}_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${P}_$name.json").read()); print("$P $name", d["value"], d["ms_per_step"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("$P $name", "FAILED", e); print(open("$O/bench_${P}_$name.err").read()[-800:])
PY
  grep "\[kx\] emit:" $O/bench_${P}_$name.err | tail -1
}
for P in apache_log csv2json iso_datetime_to_json thousand_sep; do
  run jl1 KX_JL=1; run jl0 KX_JL=0; run jl1b KX_JL=1; run jl0b KX_JL=0
done
for jl in 0 1; do
  KX_JL=$jl KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 300 python profiles/ceiling.py --kind normal --gib 2 > $O/tl_jl$jl.json 2> $O/tl_jl$jl.err; grep "emit timeline" $O/tl_jl$jl.err | tail -1
done
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "not 10gib and not rccl" > $O/pytest_jl0.txt 2>&1; echo "rc=$?" >> $O/pytest_jl0.txt; tail -3 $O/pytest_jl0.txt
