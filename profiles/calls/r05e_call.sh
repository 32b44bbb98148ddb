R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
echo "== memset probe, round-4 library"; (timeout 300 python tools/memset_capture_probe.py --lib r04) > $O/probe_r04.log 2>&1; grep -E "eager|replay" $O/probe_r04.log
echo "== memset probe, current library"; (timeout 300 python tools/memset_capture_probe.py) > $O/probe_new.log 2>&1; grep -E "eager|replay" $O/probe_new.log
echo "== determinism, current library"; (timeout 600 python tools/replay_determinism.py --videos 16) > $O/det_new_v16.log 2>&1; tail -9 $O/det_new_v16.log | cut -c1-300
(timeout 600 python tools/replay_determinism.py --videos 4) > $O/det_new_v4.log 2>&1; tail -3 $O/det_new_v4.log | cut -c1-300
echo "== bench"; (timeout 600 python bench.py --no-cpu-baseline --no-roofline) > $O/bench.log 2>&1; grep -E "timed region" $O/bench.log; grep '^{' $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('final_loss', d['config']['final_loss'])"
(time timeout 900 python -m pytest tests/test_bench_step.py tests/test_kernels_misc.py -x -q -m gpu) > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
