# Round 6, call U: the 256x256 8-wave tile without the specialised epilogue bodies compiled in (variant nofast5) vs today's library vs round 5's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O; cd $R
(CB_LIB_VARIANT=nofast5 timeout 300 python tools/infer_shapes_probe.py) 2>&1 | grep -v amdgpu | tee $O/nofast5.txt
