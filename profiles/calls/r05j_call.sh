# Round 5, call J: kernel trace of the step with the fused stem / res2 blocks
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O/trace; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -30 $O/train_step.md | cut -c1-160
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv --full > $O/step_timeline.txt 2>&1; tail -7 $O/step_timeline.txt
gzip -f $O/trace/bench_kernel_trace.csv
