#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fftconv.py -q -x -p no:cacheprovider > $O/pytest_fft.log 2>&1; rc=$?; echo "pytest fft rc=$rc" | tee $O/summary.txt
tail -12 $O/pytest_fft.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_modes.py tests/test_gpu_graphstep.py -q -x -p no:cacheprovider -k "spectranet or fused or modes or graph or captured or conv_bank" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_fftconv.py 2>&1 | grep -v amdgpu | tee $O/bench_fftconv.txt
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-graph > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench.log | cut -c1-300
