# Round 6, call Y: verification + measurement set on the tree with the grouped encoder weight gradients and the few-rows structure
R=$GRAFT_REPO_ROOT; cd $R
bash tools/final_verify.sh
bash tools/final_profiles.sh
