#!/bin/bash
# round-2 experiments: stream overlap variants
cd /root/repo; mkdir -p gpurun_out/r02x
run() { # name, env..., -- args
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline $EXTRA > gpurun_out/r02x/$name.json 2> gpurun_out/r02x/$name.log
  echo "$name: $(python -c "import json,sys; d=json.load(open('gpurun_out/r02x/$name.json')); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
}
run base A=1
run side_wgrad CB_OVERLAP_WGRAD=1
run chains2 CB_BENCH_CHAINS=2
run chains4 CB_BENCH_CHAINS=4
run chains2_side CB_BENCH_CHAINS=2 CB_OVERLAP_WGRAD=1
EXTRA="--videos 8" run half_batch A=1
EXTRA="--videos 4" run quarter_batch A=1
EXTRA="--mode tgif" run tgif_base A=1
EXTRA="--mode tgif" run tgif_chains2 CB_BENCH_CHAINS=2
