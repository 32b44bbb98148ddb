R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_res2_block.py -x -q -m gpu) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2 3; do
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_$i.log 2>&1; echo "stem with next-tile prefetch: $(grep -E 'timed region' $O/bench_$i.log | sed 's/.*done: //')"
done
