# Round 6, call AA: bias + GELU without the stored derivative as a specialised epilogue body (combination 17): the inference rows' FFN1 per tile, tests, bench rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06aa; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_kernels_gemm8.py tests/test_kernels_gemm.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
for m in 10496 5440 8832 2624; do (timeout 300 python tools/infer_shapes_probe.py $m) 2>&1 | grep -E "FFN1" | tee -a $O/probe.txt; done
for m in "--mode infer16" ""; do (timeout 300 python bench.py $m --no-cpu-baseline --no-roofline) > $O/bench.log 2>&1; echo "bench $m: $(grep -E 'timed region' $O/bench.log | sed 's/.*done: //' | head -1)"; done
