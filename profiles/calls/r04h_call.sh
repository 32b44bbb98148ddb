# Round-4 last GPU call: the bench CLI tests on the GPU after the supervisor change, smoke, one driver-style bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_zz_bench_cli.py tests/test_gemm_stream.py -x -q -m gpu) > $O/pytest_cli.log 2>&1; tail -3 $O/pytest_cli.log | head -1
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; grep smoke $O/smoke.log
(time timeout 600 python bench.py) > $O/bench.log 2>&1; grep -E "timed region|real" $O/bench.log; grep '^{' $O/bench.log > $O/bench.json
