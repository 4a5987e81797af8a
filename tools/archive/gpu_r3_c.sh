#!/bin/bash
# round 3, pass C: ring-staged window kernel (variant 5) parity + A/B + counters; depthwise kernels by rocprof
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity_modes.py -q -p no:cacheprovider -k "conv_bank_bf16x3" > $O/pytest_ring.log 2>&1; echo "pytest ring rc=$?" | tee $O/summary.txt
tail -15 $O/pytest_ring.log
timeout -k 10 300 python tools/bench_convx3.py 512 5 0,5 > $O/convx3_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/convx3_ab.txt
for V in 0 1; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/dw$V -- python3 $R/tools/bench_dwconv.py 512 $V > $R/$O/dw$V.log 2>&1); echo "dw$V rc=$?" | tee -a $O/summary.txt
  DB=$(find $O/dw$V -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 1 > $O/dwconv_v${V}_kernel_stats.csv
  rm -rf $O/dw$V
  grep -i "dwconv" $O/dwconv_v${V}_kernel_stats.csv | cut -c1-200
done
P1=$(python tools/pick_counters.py profiles/r03_rocprofv3_counters_list.txt SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE)
P2=$(python tools/pick_counters.py profiles/r03_rocprofv3_counters_list.txt SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE)
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -- python3 $R/tools/pmc_conv.py 512 3 5 > $R/$O/pmc$i.log 2>&1); echo "pmc$i rc=$?" | tee -a $O/summary.txt
done
DBS=$(find $O/pmc1 $O/pmc2 -name "*.db" 2>/dev/null | tr '\n' ' ')
python tools/rocpd_pmc_multi.py $O/r03_pmc_conv_counters_ring.json $DBS --match conv1d > $O/pmc_table.txt 2>&1
rm -rf $O/pmc1 $O/pmc2
cat $O/summary.txt
