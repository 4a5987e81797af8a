#!/bin/bash
# Round-2 GPU pass A: full GPU test-suite, the three arithmetic modes of bench.py, a kernel profile of
# the split-bf16 mode, the PCIe-inclusive line and the 2-rank one-GPU rehearsal through the self-launcher.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
python bench.py --steps 10 --warmup 3 --h2d > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 10 --warmup 3 --math bf16x3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench x3 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 4 --warmup 2 --math f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?" | tee -a $O/summary.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_x3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --math bf16x3 --no-cpu-baseline --no-branch-streams > $GRAFT_REPO_ROOT/$O/prof_x3.log 2>&1); echo "prof x3 rc=$?" | tee -a $O/summary.txt
DB=$(find $O/prof_x3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 5 > $O/x3_kernel_stats.csv
timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 64 --rehearse-one-gpu --force-branch-streams --no-cpu-baseline > $O/bench_rehearse.json 2> $O/bench_rehearse.err; echo "rehearse rc=$?" | tee -a $O/summary.txt
cut -c1-400 $O/bench_bf16.json $O/bench_x3.json $O/bench_f32.json $O/bench_rehearse.json
cat $O/summary.txt
