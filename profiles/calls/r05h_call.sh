# Round 5, call H: fused res2 bottleneck block -- GPU tests, probe (fused vs unfused, hot / cold), bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_res2_block.py tests/test_bench_step.py -x -q -m gpu) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
(timeout 300 python tools/res2_probe.py) > $O/probe.log 2>&1; grep "res2 block" $O/probe.log
for i in 1 2; do
  (CB_NO_RES2_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_unfused$i.log 2>&1; echo "unfused: $(grep -E 'timed region' $O/bench_unfused$i.log | sed 's/.*done: //')"
  (timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $O/bench_fused$i.log 2>&1; echo "fused:   $(grep -E 'timed region' $O/bench_fused$i.log | sed 's/.*done: //')"
done
