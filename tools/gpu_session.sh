#!/bin/bash
# One GPU-box session.  Usage (repo root on the GPU box): bash tools/gpu_session.sh <tag> [steps...]
# steps: wave8 pmc35 trainab hostov distbench test smoke bench benchdrv bench35 train traincpu trainprof cpab cptiles distcheck stamps b3stamps filterstamps filtersweep prof prof35 proftrain pmc pmclds filterpmc pipeablate
TAG=${1:-r01}; shift
STEPS=${@:-test smoke bench bench35 prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
has() { [[ " $STEPS " == *" $1 "* ]]; }
T0=$(date +%s)
stamp() { echo "== [$(( $(date +%s) - T0 ))s] $1"; }
if has wave8; then stamp "8-wave / 128-VGPR stream probe (VERDICT r04 item 1)"
  # (cross-compiled in the build container: cd tools/probe && hipcc --offload-arch=gfx950 -O3 -o wave8_stream_probe wave8_stream_probe.hip)
  (cd $R/tools/probe && timeout 120 ./wave8_stream_probe | tee $OUT/wave8_stream_probe.jsonl); fi
if has test; then stamp "pytest -m gpu"
  timeout 900 python -m pytest tests -q -m gpu --maxfail=8 2>&1 | tail -60 | tee $OUT/pytest_gpu.log; fi
if has smoke; then stamp smoke
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log; fi
if has bench; then stamp "bench c2"
  timeout 400 python bench.py --steps 200 --warmup 20 --details-file $OUT/bench_c2_full.json 2>&1 | tail -1 | tee $OUT/bench_c2.json; fi
if has benchdrv; then stamp "bench c2 with the driver's flags"
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --details-file $OUT/bench_c2_driverflags_full.json 2>&1 | tail -1 | tee $OUT/bench_c2_driverflags.json; fi
if has train; then stamp "train bench"
  timeout 300 python tools/train_bench.py --steps 50 2>&1 | tail -1 | tee $OUT/train_bench.json
  timeout 300 python tools/train_bench.py --steps 50 --graph 2>&1 | tail -3 | tee $OUT/train_bench_graph.json
  timeout 300 python tools/train_bench.py --steps 50 --graph --batch 512 2>&1 | tail -1 | tee -a $OUT/train_bench_graph.json
  timeout 300 python tools/train_bench.py --steps 50 --batch 512 2>&1 | tail -1 | tee -a $OUT/train_bench.json; fi
if has trainab; then stamp "train bench: weight-gradient fork on / off (graphed and eager, B = 64 and 512)"
  for b in 64 512; do for g in "--graph" ""; do for f in "--fork" "--no-fork"; do
    timeout 300 python tools/train_bench.py --steps 100 $g --batch $b $f 2>&1 | tail -1 | tee -a $OUT/train_fork_ab.jsonl
  done; done; done; fi
if has wgradab; then stamp "train bench: weight gradients merged into one launch (default) vs per layer (+ fork rule), B = 64 and 512"
  for b in 64 512; do for g in "--graph" ""; do for m in "" "--wgrad-per-layer"; do
    timeout 300 python tools/train_bench.py --steps 100 $g --batch $b $m 2>&1 | tail -1 | tee -a $OUT/train_wgrad_merged_ab.jsonl
  done; done; done; fi
if has wgradsweep; then stamp "train bench: workgroups per layer of the weight-gradient kernel (graphed, B = 64)"
  for w in 96 128 192 256 320 448; do
    timeout 300 python tools/train_bench.py --steps 100 --graph --wgrad-wgs $w 2>&1 | tail -1 | tee -a $OUT/train_wgrad_sweep.jsonl
  done; fi
if has fusedrule; then stamp "one-launch policy kernel at 1, 2, 4 rounds of the chip vs the two-kernel path"
  timeout 300 python tools/fused_rule_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fused_rule_probe.jsonl; fi
if has hostov; then stamp "host-side cost of one policy step"
  timeout 300 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids | head -70 | tee $OUT/host_overhead.txt; fi
if has shardgap; then stamp "eager 16 x 100 policy step: host cost of the pieces, device-side gaps"
  timeout 200 python tools/shard_gap_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/shard_gap_probe.jsonl
  (cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/shardgap -o trace -- python $R/tools/shard_gap_probe.py trace > $OUT/shardgap_run.log 2>&1)
  python tools/shard_gap_probe.py gaps $OUT/shardgap 2>&1 | tail -1 | tee -a $OUT/shard_gap_probe.jsonl
  find $OUT/shardgap -name "*.csv" -size +1M -delete; fi
if has traincpu; then stamp "train bench with the CPU oracle's training step beside it"
  timeout 300 python tools/train_bench.py --steps 50 --graph --cpu-seconds 8 2>&1 | tail -1 | tee $OUT/train_bench_cpu.json; fi
if has trainops; then stamp "which op launches the stray fill / copy kernels of the training step"
  timeout 300 python tools/train_host_profile.py ops 2>&1 | grep -v amdgpu.ids | head -90 | tee $OUT/train_ops.txt; fi
if has trainprof; then stamp "host profile of the eager training step"
  timeout 300 python tools/train_host_profile.py 2>&1 | grep -v amdgpu.ids | head -60 | tee $OUT/train_host_profile.txt; fi
if has distcheck; then stamp "launcher-less bench.py --gpus 2 (bench.py starts its two ranks itself; gloo, both ranks on GPU 0)"
  GNNPP_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --pmc off --dist-backend gloo --details-file $OUT/bench_selflaunch_full.json 2>&1 | tail -1 | cut -c1-700 | tee $OUT/distcheck.log
  stamp "... and without the device override: refused (one GPU here)"
  timeout 120 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --pmc off 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/distcheck.log
  stamp "2-rank gloo run of bench.py on one GPU under torch.distributed.run (code-path check only)"
  GNNPP_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --dist-backend gloo 2>&1 | tail -2 | cut -c1-600 | tee -a $OUT/distcheck.log
  stamp "2 ranks over RCCL on ONE GPU: GraphedTrainStep(dp=FlatBucketDP) -- the captured all-reduce must execute"
  GNNPP_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 tools/train_bench.py --steps 20 --graph > $OUT/rccl_one_gpu.log 2>&1; echo "exit $?" >> $OUT/rccl_one_gpu.log
  grep -i "error\|duplicate\|invalid\|agent-steps\|^exit" $OUT/rccl_one_gpu.log | grep -v amdgpu.ids | head -12 | cut -c1-400 | tee -a $OUT/distcheck.log
  stamp "1-rank torchrun nccl"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/distcheck.log; fi
if has distbench; then stamp "2-rank gloo run of bench.py on one GPU, full secondary block (rank-0-only records must not start collectives)"
  GNNPP_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29536 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --dist-backend gloo 2>&1 | tail -2 | cut -c1-700 | tee $OUT/distbench.log
  stamp "the same with --scaling strong --config c5"
  GNNPP_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --dist-backend gloo --scaling strong --config c5 2>&1 | tail -1 | cut -c1-700 | tee -a $OUT/distbench.log; fi
if has cpab; then stamp "column-packed policy kernel A/B (+ phase stamps)"
  timeout 400 python tools/cp_ab.py stamps 2>&1 | grep -v "amdgpu.ids\|warning\|note:" | tee $OUT/cp_ab.jsonl; fi
if has cptiles; then stamp "column-packed encoder tiles (latency regime) A/B + stamps"
  timeout 400 python tools/cp_ab.py stamps tiles 2>&1 | grep -v "amdgpu.ids\|warning\|note:" | tee $OUT/cp_tiles.jsonl; fi
if has filterstamps; then stamp "phase stamps of the policy filter kernels"
  timeout 300 python tools/b3_stamps.py filter 2>&1 | grep -v "amdgpu.ids\|warning\|note:" | tee $OUT/filter_stamps.jsonl; fi
if has b3stamps; then stamp "phase stamps of the policy kernels (per precision)"
  timeout 300 python tools/b3_stamps.py 2>&1 | grep -v "amdgpu.ids\|warning\|note:" | tee $OUT/b3_stamps.jsonl; fi
if has filtersweep; then stamp "filter-only throughput sweep"
  timeout 400 python tools/filter_sweep.py 2>&1 | grep -v amdgpu.ids | tee $OUT/filter_sweep.jsonl; fi
if has stamps; then stamp "phase stamps of the rollout kernels"
  timeout 300 python tools/phase_stamps.py 2>&1 | grep -v "amdgpu.ids\|warning\|note:" | tail -14 | tee $OUT/phase_stamps.jsonl; fi
if has bench35; then for c in c3 c5; do stamp "bench $c"
  timeout 400 python bench.py --config $c --steps 100 --warmup 10 --cpu-seconds 4 --details-file $OUT/bench_${c}_full.json 2>&1 | tail -1 | tee $OUT/bench_$c.json; done; fi
cd /tmp && export TMPDIR=/tmp
if has prof; then stamp "rocprofv3 kernel trace"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --pipeline-streams 0 --pmc off > $OUT/prof_run.log 2>&1
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" | tee $OUT/kernel_stats_head.csv
  find $OUT/prof -name "*kernel_trace.csv" -size +6M -delete; fi
if has prof35; then for c in c3 c5; do stamp "rocprofv3 kernel trace $c"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o trace -- python $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --pipeline-streams 0 --pmc off > $OUT/prof_run_$c.log 2>&1
  f=$(find $OUT/prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" | tee $OUT/kernel_stats_head_$c.csv
  find $OUT/prof_$c -name "*kernel_trace.csv" -size +6M -delete; done; fi
if has proftrain; then stamp "rocprofv3 kernel trace of the training step"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o trace -- python $R/tools/train_bench.py --steps 30 > $OUT/prof_run_train.log 2>&1
  f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -24 "$f" | cut -c1-200 | tee $OUT/kernel_stats_head_train.csv
  find $OUT/prof_train -name "*kernel_trace.csv" -size +6M -delete; fi
if has pmc; then stamp "rocprofv3 pmc passes"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python $R/bench.py --pmc-target > $OUT/pmc_$name.log 2>&1
    python $R/tools/pmc_summary.py $OUT/pmc_$name 2>&1 | tee -a $OUT/pmc_summary.txt
    find $OUT/pmc_$name -name "*.csv" -size +2M -delete
  done; fi
if has pmc35; then for c in c3 c5; do stamp "rocprofv3 pmc passes $c"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${c}_$name -o pmc -- python $R/bench.py --pmc-target --config $c > $OUT/pmc_${c}_$name.log 2>&1
    python $R/tools/pmc_summary.py $OUT/pmc_${c}_$name 2>&1 | tee -a $OUT/pmc_summary_$c.txt
    find $OUT/pmc_${c}_$name -name "*.csv" -size +2M -delete
  done; done; fi
if has pmclds; then stamp "rocprofv3 pmc pass: LDS conflicts of the C2 launch"
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_lds -o pmc -- python $R/bench.py --pmc-target > $OUT/pmc_lds.log 2>&1
  python $R/tools/pmc_summary.py $OUT/pmc_lds 2>&1 | grep "encoder_kernel" | tee $OUT/pmc_lds_summary.txt
  find $OUT/pmc_lds -name "*.csv" -size +2M -delete; fi
if has filterpmc; then stamp "rocprofv3 pmc passes over the filter-only launch (B = 8192 graphs of 10 nodes)"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/fpmc_$name -o pmc -- python $R/tools/filter_sweep.py --pmc-target 8192 $FILTER_PMC_ROWS > $OUT/fpmc_$name.log 2>&1
    python $R/tools/pmc_summary.py $OUT/fpmc_$name 2>&1 | grep "lsigf_" | tee -a $OUT/filter_pmc.txt
    find $OUT/fpmc_$name -name "*.csv" -size +2M -delete
  done; fi
if has pipeablate; then stamp "pipeline filter kernel: launch time and LDS conflict counters with single accesses removed"
  cd $R; timeout 300 python tools/filter_sweep.py --ablate-times 32768 2>&1 | grep -v amdgpu.ids | tee $OUT/pipe_ablate_times.jsonl
  cd /tmp
  for mask in 0 0x10 0x20 0x40 0x80 0x300 0x3f0; do
    timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/apmc_$mask -o pmc -- python $R/tools/filter_sweep.py --pmc-target 32768 64 $mask > $OUT/apmc_$mask.log 2>&1
    echo "ablate $mask" | tee -a $OUT/pipe_ablate_pmc.txt
    python $R/tools/pmc_summary.py $OUT/apmc_$mask 2>&1 | grep "lsigf_" | tee -a $OUT/pipe_ablate_pmc.txt
    find $OUT/apmc_$mask -name "*.csv" -size +2M -delete
  done; fi
stamp done
du -sh $OUT
