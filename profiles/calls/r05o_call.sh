# Round 5, call O: the other bench rows on the current tree (configs[3], configs[4], configs[2] per GPU at its native sizes) + default line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O; cd $R
(timeout 600 python bench.py) > $O/bench.log 2>&1; grep '^{' $O/bench.log > $O/bench.json; grep -E "timed region" $O/bench.log
(timeout 600 python bench.py --mode tgif --no-cpu-baseline) > $O/bench_tgif.log 2>&1; grep '^{' $O/bench_tgif.log > $O/bench_tgif.json; grep -E "timed region" $O/bench_tgif.log
(timeout 600 python bench.py --mode infer16 --no-cpu-baseline) > $O/bench_infer16.log 2>&1; grep '^{' $O/bench_infer16.log > $O/bench_infer16.json; grep -E "timed region" $O/bench_infer16.log
(timeout 600 python bench.py --size 448 --txt-len 20 --n-clips 4 --no-cpu-baseline) > $O/bench_448.log 2>&1; grep '^{' $O/bench_448.log > $O/bench_448px_c4.json; grep -E "timed region" $O/bench_448.log
