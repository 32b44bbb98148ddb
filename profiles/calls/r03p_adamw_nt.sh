# AdamW with non-temporal loads / stores of the fp32 state (CB_ADAMW_NT=1): isolated rate and the whole step, interleaved on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
for i in 1 2; do
  python tools/adamw_probe.py 2>&1 | grep cb_adamw | sed "s/^/temporal     /"
  CB_ADAMW_NT=1 python tools/adamw_probe.py 2>&1 | grep cb_adamw | sed "s/^/non-temporal /"
done | tee $O/adamw_probe.txt
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep "timed region" | sed "s/^/temporal     /"
  CB_ADAMW_NT=1 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep "timed region" | sed "s/^/non-temporal /"
done | tee $O/bench_ab.txt
