# Round-4 fifth GPU call: what makes a launch slow in the step (cold-operand breakdown); head-loss kernels on the GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
(time timeout 300 python tools/cold_breakdown.py --out $O/cold_breakdown.json) > $O/cold_breakdown.log 2>&1; grep "^{'shape" $O/cold_breakdown.log | cut -c1-420
(time timeout 600 python -m pytest tests/test_kernels_misc.py tests/test_tasks.py tests/test_clips.py -x -q -m gpu) > $O/pytest_misc.log 2>&1; tail -3 $O/pytest_misc.log | head -1
