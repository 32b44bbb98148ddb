# Same-box A/B of two launch tables: the committed csrc/gemm_tuned.h against the one generated from the cold-cache sweep
# (profiles/r03f_gemm_tuning_cold.json).  Rebuilds gemm.hip on the GPU box in between.  Run through gpurun from the repo root.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table A (committed):', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt; done
cp clipbert_amd/csrc/gemm_tuned.h $O/gemm_tuned_A.h
python tools/gen_tuned.py profiles/r02_gemm_tuning.json profiles/r03f_gemm_tuning_cold.json | tee -a $O/ab.txt
python -m clipbert_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table B (cold sweep):', d['ms_per_step'], 'ms')" | tee -a $O/ab.txt; done
for m in tgif infer16; do python bench.py --mode $m --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table B $m:', d['ms_per_step'], 'ms', d['value'])" | tee -a $O/ab.txt; done
python bench.py --size 448 --txt-len 20 --n-clips 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table B 448px c4:', d['ms_per_step'], 'ms', d['value'])" | tee -a $O/ab.txt
(timeout 600 python -m pytest tests/test_comm.py tests/test_zz_bench_cli.py tests/test_kernels_gemm8.py -m gpu -q 2>&1 | tail -6) | tee $O/pytest_part.log
