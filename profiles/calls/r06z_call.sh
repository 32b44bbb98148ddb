# Round 6, call Z: the round-end verification + measurement set (tools/final_verify.sh, tools/final_profiles.sh)
R=$GRAFT_REPO_ROOT; cd $R
bash tools/final_verify.sh
bash tools/final_profiles.sh
(timeout 600 python tools/replay_determinism.py) 2>&1 | tail -1
