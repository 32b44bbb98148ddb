# Round-4 sixth GPU call: embedding-backward slices (in the default build) and the grouped shortcut + conv1 pairs (CB_GROUP_FWD_PAIRS=1) against the default, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O/trace
cd $R
for i in 1 2 3; do
(timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_default_$i.json 2> $O/bench_default_$i.err; grep -E "timed region" $O/bench_default_$i.err
(timeout 300 env CB_GROUP_FWD_PAIRS=1 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_pairs_$i.json 2> $O/bench_pairs_$i.err; grep -E "timed region" $O/bench_pairs_$i.err
done
cd /tmp; export TMPDIR=/tmp
timeout 300 env CB_GROUP_FWD_PAIRS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -4 $O/train_step.md; grep -E "embed_bwd|group" $O/train_step.md | cut -c1-200
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv | tail -7
gzip -f $O/trace/bench_kernel_trace.csv
