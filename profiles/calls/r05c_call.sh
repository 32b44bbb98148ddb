# Round 5, call C: is the captured step deterministic from replay to replay?  new library vs the round-4 library, 4 and 16 videos
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
for v in 4 16; do
  (timeout 600 python tools/replay_determinism.py --videos $v) > $O/det_new_v$v.log 2>&1; tail -12 $O/det_new_v$v.log | cut -c1-400
  (timeout 600 python tools/replay_determinism.py --videos $v --lib r04) > $O/det_r04_v$v.log 2>&1; tail -12 $O/det_r04_v$v.log | cut -c1-400
done
(timeout 600 python tools/replay_determinism.py --videos 16 --dropout 1 --eager 0) > $O/det_new_v16_drop.log 2>&1; tail -8 $O/det_new_v16_drop.log | cut -c1-400
