# Round-4 seventh GPU call: the persistent 8-wave kernel with raw-barrier epilogue passes (schedule 4 = .../m3) against schedules 0-2 and the 4-wave kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
(time timeout 600 python -m pytest tests/test_kernels_gemm8.py -x -q -m gpu -k "persist") > $O/pytest_persist.log 2>&1; tail -2 $O/pytest_persist.log | head -1
(time timeout 900 python tools/gemm8_probe.py --quick --out $O/gemm8_probe.json) > $O/gemm8_probe.log 2>&1; grep -E "^\[probe\]|best configuration|m3" $O/gemm8_probe.log | cut -c1-260 | tail -30
