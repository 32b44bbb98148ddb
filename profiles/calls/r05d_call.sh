R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
echo "== plain"; (timeout 300 python tools/replay_debug.py) > $O/a.log 2>&1; grep -E "^replay|lazy" $O/a.log | cut -c1-330
echo "== restore"; (timeout 300 python tools/replay_debug.py --restore 1) > $O/b.log 2>&1; grep -E "^replay" $O/b.log | cut -c1-330
echo "== prezero"; (timeout 300 python tools/replay_debug.py --prezero 1) > $O/c.log 2>&1; grep -E "^replay" $O/c.log | cut -c1-330
echo "== no group wgrad"; (CB_NO_GROUP_WGRAD=1 timeout 300 python tools/replay_debug.py) > $O/d.log 2>&1; grep -E "^replay" $O/d.log | cut -c1-330
echo "== no 8w"; (CB_GEMM_NO8W=1 CB_GEMM_NO_MODEL=1 timeout 300 python tools/replay_debug.py) > $O/e.log 2>&1; grep -E "^replay" $O/e.log | cut -c1-330
