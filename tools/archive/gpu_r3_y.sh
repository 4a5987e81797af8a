#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3y; mkdir -p $O
timeout -k 10 250 python tools/dbg_coresidency4.py 2>&1 | grep "victim" | tee $O/coresidency.txt
timeout -k 10 250 python tools/dbg_coresidency5.py 2>&1 | grep "beside" | tee -a $O/coresidency.txt
for b in 8 32 128 512; do echo "B=$b: $(BATCH=$b timeout -k 10 250 python tools/dbg_graph_fwd.py 2>&1 | grep "^1s vs 1s" | cut -c1-200)" | tee -a $O/coresidency.txt; done
timeout -k 10 600 python -m pytest tests/test_gpu_fftconv.py tests/test_gpu_graphstep.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -3 $O/pytest.log
timeout -k 10 300 python tools/bench_fftconv.py 2>&1 | grep -v amdgpu | tee $O/bench_fftconv.txt
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-graph > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench.log | cut -c1-260
