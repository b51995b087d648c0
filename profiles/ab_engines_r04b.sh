#!/bin/bash
# A/B on ONE box, alternating: k_backlen with a 128-byte line (two pieces) per trip in registers (_probe/r04/libkxhip_pairload.so; it also has
# the scalar wave number in k_emit, compare k_backlen only) against one piece at a time, the next requested behind the forward sequence.
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r04ab2; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "not 10gib and not rccl" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
cp kleenexlang_amd/_build/libkxhip.so /tmp/new.so
for rep in 1 2; do
  for which in pairload new; do
    if [ $which = new ]; then cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so; else cp _probe/r04/libkxhip_$which.so kleenexlang_amd/_build/libkxhip.so; fi
    for p in apache_log csv2json iso_datetime_to_json thousand_sep; do
      timeout 600 python bench.py --program $p --steps 10 --warmup 2 --no-cpu 2>/dev/null > $O/${which}_${p}_$rep.json
      python -c "import sys,json; d=json.loads(open('$O/${which}_${p}_$rep.json').read()); print('$which', '$p', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])"
    done
  done
done
cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so
