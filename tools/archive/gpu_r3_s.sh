#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fftconv.py tests/test_gpu_graphstep.py tests/test_gpu_parity_modes.py tests/test_gpu_models.py -q -x -p no:cacheprovider -k "fft or conv_bank or spectranet or graph or captured" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt
tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_fftconv.py 2>&1 | grep -v amdgpu | tee $O/bench_fftconv.txt
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-graph > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench.log | cut -c1-300
