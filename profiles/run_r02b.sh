#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 300 ./_probe/lds_cost > $O/lds_cost.txt 2>&1
for w in 2 4; do
echo "== rccl world $w single device (NCCL_HOSTID per rank)" >> $O/rccl.txt
NCCL_DEBUG=WARN timeout 600 python bench.py --gpus $w --single-device --gib 1 --steps 3 --warmup 1 --no-cpu >> $O/rccl.txt 2>&1
echo "rc=$?" >> $O/rccl.txt
done
echo "== rccl world 1 force-dist" >> $O/rccl.txt
timeout 600 python bench.py --gpus 1 --force-dist --gib 1 --steps 3 --warmup 1 --no-cpu >> $O/rccl.txt 2>&1
echo "== gloo world 2 single device" >> $O/rccl.txt
timeout 600 python bench.py --gpus 2 --single-device --backend gloo --gib 1 --steps 3 --warmup 1 --no-cpu >> $O/rccl.txt 2>&1
echo "rc=$?" >> $O/rccl.txt
grep -E "^\{|rc=|==|Duplicate|error" $O/rccl.txt | cut -c1-300
