# Round 6, call B: LayerNorm backward -- rows-in-flight kernel (CB_LN_BWD_GEOM) vs the one-row-per-wave kernel, over the grid size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 600 python tools/ln_bwd_probe.py 2>&1 | tee $O/ln_bwd_probe.txt | tail -80
