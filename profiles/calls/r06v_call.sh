# Round 6, call V: the 256x256 8-wave forward / data-gradient kernels without the specialised epilogue bodies (permanent), QKV of the inference row on 256x128
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06v; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_kernels_gemm8.py tests/test_gemm_stream.py tests/test_gpu_full.py -m gpu -q -p no:cacheprovider) 2>&1 | tail -2
(timeout 300 python tools/infer_shapes_probe.py) 2>&1 | grep -v amdgpu | tee $O/probe_10496.txt
(timeout 300 python tools/infer_shapes_probe.py 5440) 2>&1 | grep -v amdgpu | tee $O/probe_5440.txt
(timeout 300 python tools/infer_shapes_probe.py 8832) 2>&1 | grep -v amdgpu | tee $O/probe_8832.txt
for m in "--mode infer16" "--mode tgif" "--size 448 --txt-len 20 --n-clips 4" ""; do (timeout 300 python bench.py $m --no-cpu-baseline --no-roofline) > $O/bench.log 2>&1; echo "bench $m: $(grep -E 'timed region' $O/bench.log | sed 's/.*done: //')"; done
