# Round-4: the bench CLI tests on the GPU with the hardened supervisor (done-marker grace, worker exit without teardown)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
(time timeout 1500 python -m pytest tests/test_zz_bench_cli.py -q -m gpu --durations=8) > $O/pytest_cli.log 2>&1; tail -16 $O/pytest_cli.log | cut -c1-200
