#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tail.py -q -x -p no:cacheprovider > $O/pytest_tail.log 2>&1; rc=$?; echo "pytest tail rc=$rc" | tee $O/summary.txt
tail -15 $O/pytest_tail.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_modes.py tests/test_gpu_graphstep.py -q -x -p no:cacheprovider -k "spectranet or fused or modes or graph or captured or conv_bank" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-graph > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench.log | cut -c1-300
APPLECIDER_NO_FUSED_TAIL=1 timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-fast-mode --no-h2d --no-ceilings > $O/bench_unfused.log 2>&1
tail -1 $O/bench_unfused.log | cut -c1-300
