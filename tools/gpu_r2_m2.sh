#!/bin/bash
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
timeout -k 10 300 python -X faulthandler bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode > $O/b2.json 2> $O/b2.err; echo "bench rc=$?"
tail -60 $O/b2.err
