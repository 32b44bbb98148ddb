# Round-4 diagnostic (NOT the metric): the same step at 2x / 4x the videos per GPU -- how much of the 9.5 ms is batch-independent
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
for v in 16 32 64; do
(timeout 300 python bench.py --videos $v --no-cpu-baseline --no-roofline) > $O/bench_v$v.json 2> $O/bench_v$v.err; echo "videos $v: $(grep -E 'timed region' $O/bench_v$v.err)"
done
