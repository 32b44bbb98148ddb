R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gemm_group.py tests/test_bench_step.py -x -q -m gpu) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
(CB_GEMM_NO_STREAMK=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_splitk1.log 2>&1; echo "split-K groups: $(grep -E 'timed region' $O/bench_splitk1.log | sed 's/.*done: //')"
for c in 1.0 0.5 2.0 0.25; do
  (CB_GEMM_STREAMK_CHUNK=$c timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_sk_$c.log 2>&1; echo "stream-K chunk $c: $(grep -E 'timed region' $O/bench_sk_$c.log | sed 's/.*done: //')"
done
(CB_GEMM_NO_STREAMK=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_splitk2.log 2>&1; echo "split-K groups: $(grep -E 'timed region' $O/bench_splitk2.log | sed 's/.*done: //')"
