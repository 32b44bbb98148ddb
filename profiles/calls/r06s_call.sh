# Round 6, call S: the inference row's encoder products per tile, specialised vs generic epilogue
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
(timeout 300 python tools/infer_shapes_probe.py) 2>&1 | grep -v amdgpu | tee $O/fast.txt
(CB_GEMM_FAST_EPI=0 timeout 300 python tools/infer_shapes_probe.py) 2>&1 | grep -v amdgpu | tee $O/generic.txt
