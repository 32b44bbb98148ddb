# kernel trace + stats of the bench command on the final HEAD
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z_head; mkdir -p $O/trace
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -8 $O/train_step.md
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv > $O/step_phases.txt 2>&1; cat $O/step_phases.txt
gzip -f $O/trace/bench_kernel_trace.csv; tail -1 $O/trace/bench.log | cut -c1-400
