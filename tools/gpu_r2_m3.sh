#!/bin/bash
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
rm -f $O/probe.log
for cfg in "bf16x3 512 1" "bf16x3 256 1"; do
  echo "== $cfg" >> $O/probe.log
  timeout -k 10 200 python -X faulthandler tools/exp_graph_step.py $cfg >> $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
done
echo "== bench no-branch-streams" >> $O/probe.log
timeout -k 10 300 python -X faulthandler bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams > $O/b3.json 2>> $O/probe.log; echo "rc=$?" >> $O/probe.log
grep -v "amdgpu.ids\|Extension modules\|UserWarning\|Consider using\|print(mode" $O/probe.log | tail -80
