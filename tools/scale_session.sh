#!/bin/bash
# 1 -> 8 GPU scaling session on ONE node (run from the repo root on an 8-GPU MI355X box):
#   bash tools/scale_session.sh [tag]
# Writes one JSON line per GPU count to gpurun_out/<tag>/scale_policy.jsonl (bench.py: rollout replicas,
# weak scaling, no data-path collective), scale_policy_strong_c5.jsonl (bench.py --config c5 --scaling strong: the
# 128-graph batch of config 5 sharded over the ranks, 16 graphs per GPU at N = 8) and scale_train.jsonl (tools/train_bench.py: config 4, FlatBucketDP =
# one flat-bucket all-reduce per step over RCCL / xGMI), each line carrying the rank count RCCL reports.
# The driver computes scaling efficiency from the per-N values itself; nothing here reports one.
TAG=${1:-scale}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NGPU" | tee $OUT/scale_info.txt
: > $OUT/scale_policy.jsonl; : > $OUT/scale_train.jsonl; : > $OUT/scale_policy_strong_c5.jsonl
PORT=29600
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "skipping N=$N (only $NGPU GPUs)" | tee -a $OUT/scale_info.txt; continue; }
  PORT=$((PORT + 1))
  if [ "$N" -eq 1 ]; then
    timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --pmc off 2>/dev/null | tail -1 >> $OUT/scale_policy.jsonl
    timeout 600 python tools/train_bench.py --steps 100 2>/dev/null | tail -1 >> $OUT/scale_train.jsonl
    timeout 600 python bench.py --gpus 1 --config c5 --scaling strong --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --pmc off 2>/dev/null | tail -1 >> $OUT/scale_policy_strong_c5.jsonl
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port $PORT bench.py --gpus $N --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --pmc off \
      2>/dev/null | tail -1 >> $OUT/scale_policy.jsonl
    PORT=$((PORT + 1))
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port $PORT tools/train_bench.py --steps 100 2>/dev/null | tail -1 >> $OUT/scale_train.jsonl
    PORT=$((PORT + 1))
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port $PORT bench.py --gpus $N --config c5 --scaling strong --steps 100 --warmup 10 --no-cpu-baseline \
      --no-secondary --pmc off 2>/dev/null | tail -1 >> $OUT/scale_policy_strong_c5.jsonl
  fi
done
echo "== policy forward (agent-steps/s, whole job)"; cut -c1-220 $OUT/scale_policy.jsonl
echo "== policy forward, config 5 strong scaling (agent-steps/s, whole job)"; cut -c1-220 $OUT/scale_policy_strong_c5.jsonl
echo "== training (agent-steps/s, whole job)"; cut -c1-400 $OUT/scale_train.jsonl
# fail loudly unless every N-GPU line really used N ranks on N distinct GPUs, and N = 1 matches the committed line
RC=0
python tools/check_scale.py $OUT/scale_policy.jsonl --reference profiles/r03_bench_c2.json | tee $OUT/scale_check.txt || RC=1
python tools/check_scale.py $OUT/scale_train.jsonl | tee -a $OUT/scale_check.txt || RC=1
python tools/check_scale.py $OUT/scale_policy_strong_c5.jsonl | tee -a $OUT/scale_check.txt || RC=1
[ $RC -ne 0 ] && echo "!! scaling session INVALID (see messages above)"
exit $RC
