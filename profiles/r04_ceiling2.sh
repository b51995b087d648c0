#!/bin/bash
# the two ceiling experiments of r04a once more, on the FINAL engine of round 4 (operands prefetched: does the LDS matter more now?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
ulimit -c 0
for rep in 1 2; do for kind in normal p256; do
  timeout 300 python profiles/ceiling.py --kind $kind --gib 2 > $O/plain_${kind}_$rep.json 2> /dev/null; tail -1 $O/plain_${kind}_$rep.json | cut -c1-330
done
KX_DEBUG_FLAGS=32 timeout 300 python profiles/ceiling.py --kind normal --gib 2 > $O/pitch_$rep.json 2> /dev/null; tail -1 $O/pitch_$rep.json | cut -c1-330
done
