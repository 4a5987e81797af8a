#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-branch-streams --no-h2d --no-ceilings --no-graph > $R/$O/prof.log 2>&1); echo "stats rc=$?"
DB=$(find $O/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 8 > $O/kernel_stats.csv
rm -rf $O/prof
grep -E "ln_gelu_pw|layernorm_fwd_sub_kernelILi16|maxpool4_fwd|layernorm_bwd_sub_kernelILi16" $O/kernel_stats.csv | cut -c1-200 | sort -u
