#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_modes.py tests/test_gpu_models.py -q -x -p no:cacheprovider -k "gemm or linear or split or fused or spectranet or astrominn or wgrad" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/summary.txt
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-graph > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench.log | cut -c1-300
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR -d $R/$O/pmc_$CTR -- python3 $R/bench.py --steps 1 --warmup 1 --math bf16x3 --no-cpu-baseline --no-fast-mode --no-graph --no-branch-streams --no-h2d --no-ceilings > $R/$O/pmc_$CTR.log 2>&1); echo "pmc $CTR rc=$?" | tee -a $O/summary.txt
  DB=$(find $O/pmc_$CTR -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py $DB $CTR > $O/pmc_$CTR.json
  rm -rf $O/pmc_$CTR
done
python tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json > $O/pmc_hbm_traffic_bf16x3.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3t/pmc_hbm_traffic_bf16x3.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'gemm_x3' in k: print(k[-40:], v)
PY
