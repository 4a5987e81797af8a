#!/bin/bash
mkdir -p gpurun_out/r2u
O=gpurun_out/r2u
APPLECIDER_TRACE_STACKS=4 timeout -k 10 400 python bench.py --gpus 2 --rehearse-one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --h2d > $O/a.json 2> $O/a.err; echo "h2d-only rc=$?"
grep -v "^\[Gloo\]\|amdgpu" $O/a.err | head -120
