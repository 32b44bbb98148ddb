# Round 5, call R: what bounds the epilogue?  In-kernel stamps of the captured step on four diagnostic builds / settings:
#   wt1 (product settings), wt0 (write-back stores), nostore (all epilogue instructions, no global stores), nomath (raw accumulators stored)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R
(timeout 300 python tools/stamps_run.py --out $O/wt1) > $O/wt1.log 2>&1; tail -1 $O/wt1.log | cut -c1-100
(CB_GEMM_WT=0 timeout 300 python tools/stamps_run.py --out $O/wt0) > $O/wt0.log 2>&1
(timeout 300 python tools/stamps_run.py --lib stamps_nostore --out $O/nostore) > $O/nostore.log 2>&1
(timeout 300 python tools/stamps_run.py --lib stamps_nomath --out $O/nomath) > $O/nomath.log 2>&1
for v in wt1 wt0 nostore nomath; do echo "== $v: $(head -1 $O/$v/stamps.md)"; grep -E "2624x(2304|3072|768)x(768|3072|2304) " $O/$v/stamps.md | head -9 | cut -c1-200; done
rm -f $O/*/stamps_raw.npz
