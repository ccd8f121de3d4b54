#!/bin/bash
# Copy the judged evidence of ONE gpu_session.sh run (stages: test smoke bench benchdrv bench35 prof prof35 proftrain train
# traincpu pmc pmc35 filterstamps) from gpurun_out/<tag>/ into profiles/<round>_*.  Usage: bash tools/collect_profiles.sh r06i r06
TAG=${1:?session tag}; R=${2:?round prefix}
G=gpurun_out/$TAG; P=profiles
set -e
for c in c2 c2_driverflags c3 c5; do cp $G/bench_$c.json $P/${R}_bench_$c.json; cp $G/bench_${c}_full.json $P/${R}_bench_${c}_full.json; done
cp $G/kernel_stats_head.csv $P/${R}_c2_kernel_stats.csv; cp $G/kernel_stats_head_c3.csv $P/${R}_c3_kernel_stats.csv; cp $G/kernel_stats_head_c5.csv $P/${R}_c5_kernel_stats.csv
cp $G/pmc_summary.txt $P/${R}_c2_pmc_summary.txt; cp $G/pmc_summary_c3.txt $P/${R}_c3_pmc_summary.txt; cp $G/pmc_summary_c5.txt $P/${R}_c5_pmc_summary.txt
cp $G/prof_train/trace_kernel_stats.csv $P/${R}_train_kernel_stats.csv
cat $G/train_bench.json $G/train_bench_graph.json $G/train_bench_cpu.json | grep "^{" > $P/${R}_train_bench.jsonl
grep "^{" $G/filter_stamps.jsonl > $P/${R}_filter_stamps.jsonl
cp $G/pytest_gpu.log $P/${R}_pytest_gpu.log; cat $G/smoke.log >> $P/${R}_pytest_gpu.log
python - "$G" "$P/${R}_train_kernel_trace_step.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + '/prof_train/trace_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
out = ['# one steady-state EAGER training step (64 x 10, tools/train_bench.py under rocprofv3 --kernel-trace): start us, duration us, kernel',
       '# (eager: the gaps are host time; a HIP-graph replay runs the same kernels back to back)']
tot = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.append('%8.1f %6.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d,
                                    r['Kernel_Name'].replace('void ', '').replace('gnnpp::', '').split('(')[0][:60]))
out.append('# %d launches, %.1f us of kernels' % (b - a, tot))
open(sys.argv[2], 'w').write('\n'.join(out) + '\n')
print(out[-1])
PY
ls -la $P | grep " ${R}_" | awk '{print $5, $9}'
