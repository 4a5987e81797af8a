#!/bin/bash
# SQ counters of gemm_x3_kernel at two workgroups per CU.  usage: gpurun -- "bash tools/gpu_x3_pmc.sh r4p"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-x3pmc}; mkdir -p $O
R=$GRAFT_REPO_ROOT
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -- python3 $R/tools/pmc_x3.py > $R/$O/pmc$i.log 2>&1); echo "pmc$i rc=$?"
  DB=$(find $O/pmc$i -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc_multi.py $O/pmc$i.json $DB --match gemm_x3 > $O/pmc$i.out 2>&1
  rm -rf $O/pmc$i
  tail -30 $O/pmc$i.out | cut -c1-200
done
