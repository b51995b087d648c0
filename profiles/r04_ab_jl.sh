#!/bin/bash
# round 4: job-stride layout (KX_JL=1, default) against the vote-and-rank sweep (KX_JL=0), same box, alternating
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04b}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "not 10gib and not rccl and not binary" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for p in ${PROGS:-apache_log csv2json iso_datetime_to_json thousand_sep}; do for jl in 0 1 0 1; do
  KX_JL=$jl KX_DEBUG=1 timeout 600 python bench.py --program $p --steps 5 --warmup 1 --no-cpu > $O/bench_${p}_jl$jl.json 2> $O/bench_${p}_jl$jl.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${p}_jl$jl.json").read()); print("$p jl=$jl", d["value"], d["ms_per_step"], d["kernels_ms"], d["output_checked_bit_exact"])
except Exception as e: print("$p jl=$jl", "FAILED", e); print(open("$O/bench_${p}_jl$jl.err").read()[-800:])
PY
  grep "\[kx\] emit:" $O/bench_${p}_jl$jl.err | tail -1
done; done
for jl in 0 1; do
  KX_JL=$jl KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 300 python profiles/ceiling.py --kind normal --gib 2 > $O/tl_jl$jl.json 2> $O/tl_jl$jl.err; tail -1 $O/tl_jl$jl.json; grep "emit timeline" $O/tl_jl$jl.err | tail -1
done
