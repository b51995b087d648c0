#!/bin/bash
# Round-4 evidence of the final engine (run on the GPU box through gpurun): GPU test suite, the three 10 GiB bench lines,
# rocprofv3 kernel stats + HBM traffic (collect.sh), SQ counters (collect_sq.sh), soak.  usage: profiles/run_final_r03.sh TAG
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04z}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_pytest.txt 2>&1; tail -2 $O/gpu_pytest.txt
for p in apache_log csv2json iso_datetime_to_json; do
  timeout 900 python bench.py --program $p --steps 10 --warmup 2 $([ $p = apache_log ] || echo --no-cpu) > $O/bench_$p.json 2> $O/bench_$p.err
  python -c "import json; d=json.loads(open('$O/bench_$p.json').read()); print('$p', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])"
done
bash profiles/collect.sh $TAG > $O/collect.log 2>&1; tail -3 $O/collect.log
bash profiles/collect_sq.sh ${TAG}_sq > $O/collect_sq.log 2>&1; tail -3 $O/collect_sq.log
SOAK_LO=5000 SOAK_HI=6500 timeout 1500 python tests/soak/soak_engine.py > $O/soak_engine.txt 2>&1; tail -2 $O/soak_engine.txt
timeout 1200 python tests/soak/soak_windows.py > $O/soak_windows.txt 2>&1; tail -1 $O/soak_windows.txt
timeout 600 python bench.py --program thousand_sep --steps 10 --warmup 2 --no-cpu > $O/bench_thousand_sep.json 2>/dev/null; cut -c1-200 $O/bench_thousand_sep.json
timeout 600 python profiles/coder_bench.py 4 2>/dev/null | tail -1 > $O/coder_bench_csv_rows_4gib.json; cut -c100-600 $O/coder_bench_csv_rows_4gib.json
# the job-stride layout (opt-in): parity tests and the three bench lines with KX_JL=1
KX_JL=1 timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "not 10gib and not rccl" > $O/gpu_pytest_jl1.txt 2>&1; tail -1 $O/gpu_pytest_jl1.txt
for p in apache_log csv2json iso_datetime_to_json; do KX_JL=1 timeout 900 python bench.py --program $p --steps 10 --warmup 2 --no-cpu > $O/bench_jl1_$p.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench_jl1_$p.json').read()); print('jl1 $p', d['value'], d['kernels_ms'], d['output_checked_bit_exact'])"; done
KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 300 python profiles/ceiling.py --kind normal --gib 2 > $O/timeline.json 2> $O/timeline.err; grep "emit timeline" $O/timeline.err | tail -1 > $O/timeline.txt; cat $O/timeline.txt
python profiles/actions_bench.py > $O/actions_16m.json 2>/dev/null; KX_BENCH_MIB=1024 python profiles/actions_bench.py > $O/actions_1g.json 2>/dev/null; tail -c 400 $O/actions_1g.json
