#!/bin/bash
# Round 6, third run on one box: the exact slow path.  Clean log / 100 and 1000 escaped quotes in 10 GiB (lane-local resolution) / the same with
# round 5's whole-shard fall-back (KX_NO_SLOW=1, back-off off so that every step pays it)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_ab3.txt
: > $OUT
run() {  # label env... -- args
  label=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu $ARGS 2>>gpurun_out/r06_ab3.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['kernels_ms'], d['output_checked_bit_exact'], d.get('escaped_quotes_injected'), d.get('delayed_form_state_after'), d.get('clocks',{}).get('after'))" >> $OUT
}
for rep in 1 2; do
  ARGS="--program apache_log" run clean X=1
  ARGS="--program apache_log --escapes 100" run esc100 X=1
  ARGS="--program apache_log --escapes 1000" run esc1000 X=1
done
ARGS="--program apache_log --escapes 100" run esc100_noslow KX_NO_SLOW=1 KX_DF_BACKOFF_OFF=1
ARGS="--program csv2json" run csv X=1
ARGS="--program iso_datetime_to_json" run iso X=1
cat $OUT
KX_DEBUG=1 python bench.py --program apache_log --escapes 100 --steps 2 --warmup 1 --no-cpu 2>&1 >/dev/null | grep "slow path" | sort | uniq -c | head -5 >> $OUT
cat $OUT
