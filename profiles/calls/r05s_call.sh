# Round 5, call S: per-workgroup stamp records of the captured step (raw), for the tail analysis of the 64x64-tile launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R
(timeout 300 python tools/stamps_run.py --out $O/wt1) > $O/wt1.log 2>&1; tail -1 $O/wt1.log | cut -c1-100; ls -la $O/wt1
