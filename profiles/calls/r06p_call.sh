# Round 6, call P: in-kernel stamps (diagnostic build) of the captured step: the grouped encoder weight gradients' prologue / K loop / epilogue / drain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
(timeout 900 python tools/stamps_run.py --out $O) > $O/stamps.log 2>&1; head -3 $O/stamps.log; grep -E "b12" $O/stamps.md | head -12
