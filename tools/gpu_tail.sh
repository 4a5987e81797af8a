#!/bin/bash
# tail kernel tuning loop: tests + microbench (+ counters with "pmc" as $2).  usage: gpurun -- "bash tools/gpu_tail.sh r4f [pmc]"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-tail}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_tail.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/bench_tail.py 20 2>&1 | grep -v amdgpu.ids | tee $O/bench_tail.txt
if [ "$2" = "pmc" ]; then
  i=0
  for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -- python3 $R/tools/bench_tail.py 2 > $R/$O/pmc$i.log 2>&1); echo "pmc$i rc=$?"
    DB=$(find $O/pmc$i -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc_multi.py $O/pmc$i.json $DB --match tail_ > $O/pmc$i.out 2>&1
    rm -rf $O/pmc$i
  done
fi
