#!/bin/bash
# re-tune the launch table with the round's final kernels and A/B it in the same call (same box)
cd /root/repo; O=gpurun_out/r02x; mkdir -p $O
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
echo "old table train: $(b) | $(b)"; echo "old table tgif: $(b --mode tgif)"; echo "old table infer16: $(b --mode infer16)"
(time timeout 900 python tools/tune_gemm.py --modes "train;tgif;infer16;train:--size 448 --txt-len 20 --n-clips 4" --out $O/gemm_tuning_v3.json) > $O/tune_v3.log 2>&1; tail -2 $O/tune_v3.log
cp clipbert_amd/csrc/gemm_tuned.h $O/gemm_tuned_old.h
python tools/gen_tuned.py $O/gemm_tuning_v3.json 2>&1 | tail -2
cp clipbert_amd/csrc/gemm_tuned.h $O/gemm_tuned_v3.h
(time python -m clipbert_amd.build) 2>&1 | tail -4
echo "new table train: $(b) | $(b)"; echo "new table tgif: $(b --mode tgif)"; echo "new table infer16: $(b --mode infer16)"
