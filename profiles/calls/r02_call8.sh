#!/bin/bash
# what data parallelism costs a rank besides link time: the N-rank replay plan with the collectives skipped, vs the 1-GPU plan (same box)
cd /root/repo; O=gpurun_out/r02x; mkdir -p $O
for v in 0 8 0 8; do
  CB_BENCH_DRY_DP=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2> $O/dry_dp_$v.log > $O/dry_dp_$v.json
  python -c "import json; d=json.load(open('$O/dry_dp_$v.json')); print('dry_dp=$v', d['ms_per_step'], d['value'], d['config']['parallelism'], '|', d['config']['replay_plan'][:60])"
done
