# Round-4: the GPU suite, smoke and the driver-style bench line on the very last HEAD
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z_head; mkdir -p $O
cd $R
rm -f $R/gpurun_out/r04_bf16_parity.json
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; grep smoke $O/smoke.log
(time timeout 600 python bench.py) > $O/bench.log 2>&1; grep -E "timed region|real" $O/bench.log; grep '^{' $O/bench.log > $O/bench.json
