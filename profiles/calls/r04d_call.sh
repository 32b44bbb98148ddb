# Round-4 fourth GPU call: streaming variants as shipped (64x256 K<=64, 64x128 K<=128): parity, cold probe, step A/B; then the whole GPU suite on this tree.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
(time timeout 300 python -m pytest tests/test_gemm_stream.py tests/test_gemm_plan.py -x -q -m gpu) > $O/pytest_stream.log 2>&1; tail -2 $O/pytest_stream.log | head -1
(time timeout 600 python tools/stream_probe.py --out $O/stream_probe.json) > $O/stream_probe.log 2>&1; grep -E "^\{'form|^\{'problems" $O/stream_probe.log | cut -c1-330
for i in 1 2 3; do
(timeout 300 env CB_GEMM_NO_STREAM=1 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_nostream_$i.json 2> $O/bench_nostream_$i.err; grep -E "timed region" $O/bench_nostream_$i.err
(timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_stream_$i.json 2> $O/bench_stream_$i.err; grep -E "timed region" $O/bench_stream_$i.err
done
rm -f $R/gpurun_out/r04_bf16_parity.json
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
