#!/bin/bash
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_graphstep.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_parity_modes.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_stats.py $DB 8 > $O/x3_kernel_stats.csv; rm -rf $O/prof
grep "colsum\|act_bwd\|dropout\|TOTAL\|add" $O/x3_kernel_stats.csv | cut -c1-160
