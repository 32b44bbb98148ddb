R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_res2_block.py -x -q -m gpu) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do
  (CB_STEM_4WAVE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_4w$i.log 2>&1; echo "stem 4 waves: $(grep -E 'timed region' $O/bench_4w$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_8w$i.log 2>&1; echo "stem 8 waves: $(grep -E 'timed region' $O/bench_8w$i.log | sed 's/.*done: //')"
done
