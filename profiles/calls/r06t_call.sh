# Round 6, call T: the inference row's encoder products on the round-5 library (variant built from commit 9db4b00's csrc) next to today's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O; cd $R
(CB_LIB_VARIANT=r05 timeout 300 python tools/infer_shapes_probe.py) 2>&1 | grep -v amdgpu | tee $O/r05.txt
