#!/bin/bash
# The round-start engine: git show 2e11768:kleenexlang_amd/csrc/engine/{kx_engine.hip,kx_sweeps.inc,kx_sharded.cpp} + include/ built with the hipcc line of build.py into _probe/old/libkxhip.so
# A/B on one box: current engine vs the round-start engine (_probe/old/libkxhip.so), three programs, alternating
cd ${GRAFT_REPO_ROOT:-.}
cp kleenexlang_amd/_build/libkxhip.so /tmp/new.so
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp _probe/old/libkxhip.so kleenexlang_amd/_build/libkxhip.so; else cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so; fi
    for p in apache_log csv2json iso_datetime_to_json; do
      python bench.py --program $p --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', '$p', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])"
    done
  done
done
cp /tmp/new.so kleenexlang_amd/_build/libkxhip.so
for inl in 0 1; do
  KX_INL=$inl python bench.py --program thousand_sep --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('thousand_sep KX_INL=$inl', d['value'], d['ms_per_step'], d['kernels_ms'], d['output_checked_bit_exact'])"
done
