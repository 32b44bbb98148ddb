# Round 5, call T: tail probe (second workgroup of a CU) of one encoder product in several surroundings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R
(timeout 300 python tools/tail_probe.py) > $O/tail_probe.txt 2>&1; cat $O/tail_probe.txt | cut -c1-330
