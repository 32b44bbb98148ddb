# Round 6, call AC: launch table of the 448 px x 4 clips row judged by its captured step (tools/tune_instep.py; mode key of the sweep JSON through CB_BENCH_TUNE_MODE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ac; mkdir -p $O; cd $R
(CB_BENCH_TUNE=$O/instep_448c4.json CB_BENCH_TUNE_MODE="train:--size 448 --txt-len 20 --n-clips 4" CB_BENCH_TUNE_SHAPES=30 timeout 2400 python bench.py --size 448 --txt-len 20 --n-clips 4 --no-cpu-baseline --no-roofline) > $O/tune.log 2>&1
grep -E "^\[instep\]" $O/tune.log | grep -E "KEEP|baseline|overrides" | cut -c1-330 | tail -30
