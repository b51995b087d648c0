#!/bin/bash
# A/B on ONE box, alternating: the round-3 final engine (git show 83f324d:kleenexlang_amd/csrc/engine/* built with build.py's hipcc line into
# _probe/r04/libkxhip_r03final.so), the current engine with the wave number as a vector value (_probe/r04/libkxhip_vecwave.so), the current engine.
# Same compiler output (blobs), same bench.py, every output byte checked.
ulimit -c 0; export HSA_COREDUMP_PATTERN=/dev/null
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r04ab; mkdir -p $O
cp kleenexlang_amd/_build/libkxhip.so /tmp/new.so
for rep in 1 2; do
  for which in r03final vecwave new; do
    if [ $which = new ]; then cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so; else cp _probe/r04/libkxhip_$which.so kleenexlang_amd/_build/libkxhip.so; fi
    for p in apache_log csv2json iso_datetime_to_json thousand_sep; do
      timeout 600 python bench.py --program $p --steps 10 --warmup 2 --no-cpu 2>/dev/null > $O/${which}_${p}_$rep.json
      python -c "import sys,json; d=json.loads(open('$O/${which}_${p}_$rep.json').read()); print('$which', '$p', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])"
    done
  done
done
cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so
