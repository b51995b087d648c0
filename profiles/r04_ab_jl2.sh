#!/bin/bash
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04c}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_pipeline_seam.py tests/test_regex_coder.py -m gpu -x -q -k "not 10gib and not rccl" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
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
for P in apache_log csv2json iso_datetime_to_json; do
  run jl0 KX_JL=0; run jl1 KX_JL=1; run jl0b KX_JL=0; run jl1b KX_JL=1
done
P=apache_log
for s in 13 17; do run js$s KX_JL=1 KX_JSLOTS=$s; done
for jl in 0 1; do
  KX_JL=$jl KX_DEBUG=1 KX_DEBUG_FLAGS=64 timeout 300 python profiles/ceiling.py --kind normal --gib 2 > $O/tl_jl$jl.json 2> $O/tl_jl$jl.err; grep "emit timeline" $O/tl_jl$jl.err | tail -1
done
cd /tmp && export TMPDIR=/tmp
for jl in 0 1; do for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY; do
  rm -rf /tmp/pmc_$c
  KX_JL=$jl timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/profiles/ceiling.py --kind normal --gib 2 --steps 1 > /tmp/pmc_$c.log 2>&1
  python3 - $c jl$jl $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) <<'PY' | tee -a $O/sq_jl$jl.txt
import csv, re, sys, collections
c, kind, f = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    m = re.search(r"k_\w+", r["Kernel_Name"])
    if m and r["Counter_Name"] == c:
        acc[m.group(0)][0] += 1; acc[m.group(0)][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    if k in ("k_emit",): print(kind, c, k, v / n, "per 4KiB-iteration", v / n / (2 * 2**30 / 4096))
PY
done; done
