#!/bin/bash
# round 3, pass B: default bench line; MFMA-busy / LDS / clock counter passes on the conv kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "dwconv" > $O/pytest_dwconv.log 2>&1; echo "pytest dwconv rc=$?" | tee $O/summary.txt
tail -5 $O/pytest_dwconv.log
timeout -k 10 120 python tools/bench_dwconv.py 512 0 > $O/dwconv_v0.txt 2>&1; timeout -k 10 120 python tools/bench_dwconv.py 512 1 > $O/dwconv_v1.txt 2>&1
cat $O/dwconv_v0.txt $O/dwconv_v1.txt
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
(cd /tmp && rocprofv3 -L > $R/$O/counters_list.txt 2>&1); echo "list rc=$?" | tee -a $O/summary.txt
P1=$(python tools/pick_counters.py $O/counters_list.txt SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT)
P2=$(python tools/pick_counters.py $O/counters_list.txt SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE)
P3=$(python tools/pick_counters.py $O/counters_list.txt SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE)
echo "P1=$P1" | tee -a $O/summary.txt; echo "P2=$P2" | tee -a $O/summary.txt; echo "P3=$P3" | tee -a $O/summary.txt
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  [ -z "$P" ] && continue
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d $R/$O/pmc$i -- python3 $R/tools/pmc_conv.py 512 3 > $R/$O/pmc$i.log 2>&1); echo "pmc$i rc=$?" | tee -a $O/summary.txt
done
DBS=$(find $O/pmc1 $O/pmc2 $O/pmc3 -name "*.db" 2>/dev/null | tr '\n' ' ')
python tools/rocpd_pmc_multi.py $O/r03_pmc_conv_counters.json $DBS --match conv1d,pad_rows,split > $O/pmc_table.txt 2>&1
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
grep -i "mfma\|lds" $O/counters_list.txt | head -60 > $O/counters_mfma_lds.txt
cat $O/pmc_table.txt | head -120
cut -c1-1500 $O/bench_default.json
cat $O/summary.txt
