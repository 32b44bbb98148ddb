# Round-end verification set (run through gpurun from the repo root); outputs in gpurun_out/final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O/trace
cd $R
(time timeout 900 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; grep smoke $O/smoke.log
(time timeout 600 python bench.py) > $O/bench.log 2>&1; grep -E "timed region|real" $O/bench.log


cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace/bench.log 2>&1
grep -c . $O/trace/bench_kernel_stats.csv
