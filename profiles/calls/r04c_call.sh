# Round-4 third GPU call: the streaming structure (cb_gemm tile 8) -- parity on the GPU, per-problem cold-cache probe, step A/B, kernel trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O/trace
cd $R
(time timeout 600 python -m pytest tests/test_gemm_stream.py tests/test_kernels_gemm.py tests/test_model_small.py -x -q -m gpu) > $O/pytest_stream.log 2>&1; tail -3 $O/pytest_stream.log
(time timeout 600 python tools/stream_probe.py --out $O/stream_probe.json) > $O/stream_probe.log 2>&1; grep -E "^\{'form|^\{'problems" $O/stream_probe.log | cut -c1-330
for i in 1 2; do
(timeout 300 env CB_GEMM_NO_STREAM=1 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_nostream_$i.json 2> $O/bench_nostream_$i.err; grep -E "timed region" $O/bench_nostream_$i.err
(timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_stream_$i.json 2> $O/bench_stream_$i.err; grep -E "timed region" $O/bench_stream_$i.err
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace/bench.log 2>&1
python $R/tools/trace_summary.py $O/trace/bench_kernel_trace.csv > $O/train_step.md 2>&1; head -14 $O/train_step.md
python $R/tools/step_timeline.py $O/trace/bench_kernel_trace.csv | tail -7
gzip -f $O/trace/bench_kernel_trace.csv
