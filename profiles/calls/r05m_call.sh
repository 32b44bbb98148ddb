# Round 5, call M: stream-K grouped weight gradients -- GPU tests, bench A/B, the moved diagnostic modes still run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gemm_group.py tests/test_bench_step.py tests/test_model_small.py -x -q -m gpu) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do
  (CB_GEMM_NO_STREAMK=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_splitk$i.log 2>&1; echo "split-K groups: $(grep -E 'timed region' $O/bench_splitk$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_streamk$i.log 2>&1; echo "stream-K groups: $(grep -E 'timed region' $O/bench_streamk$i.log | sed 's/.*done: //')"
done
(CB_BENCH_CHAINS=2 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2) > $O/bench_chains.log 2>&1; echo "chains=2: $(grep -E 'timed region' $O/bench_chains.log | sed 's/.*done: //')"
(CB_BENCH_PIPELINE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2) > $O/bench_pipe.log 2>&1; echo "pipeline: $(grep -E 'timed region' $O/bench_pipe.log | sed 's/.*done: //')"
