# Round 6, call N (experiment): ordered slab K split on forward products with a full epilogue
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
(CB_EXP_SLAB_ANY=1 timeout 600 python tools/slab_fwd_probe.py) 2>&1 | tee $O/slab_fwd_probe.txt | tail -40
