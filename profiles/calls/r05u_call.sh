# Round 5, call U: unstamped per-launch time of the N = 768 encoder products by tile (graph of 48 launches)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
(timeout 300 python tools/tile_time_probe.py) > $O/tile_time.txt 2>&1; cat $O/tile_time.txt | cut -c1-250
