#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace (+ optional PMC passes).
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh [tag] [pmc]
TAG=${1:-r01}
PMC=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
echo "== build check"; ls -la gnn_pathplanning_amd/libgnnpp.so
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench c2"
timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee $OUT/bench_c2.json
for c in c3 c5; do
  echo "== bench $c"
  timeout 600 python bench.py --config $c --steps 100 --warmup 10 --cpu-seconds 4 2>&1 | tail -1 | tee $OUT/bench_$c.json
done
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/prof_run.log 2>&1
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | tee $OUT/kernel_stats_head.csv
if [ -n "$PMC" ]; then
  echo "== rocprofv3 pmc passes"
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
  ls $OUT/pmc_*/ 2>/dev/null | head
fi
# keep the merged-back payload small
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
