# Same-box A/B of CB_FWD_EPF (csrc/gemm_impl.h): rebuilds the 64x64 / 128x64 4-wave instantiations with the forward residual prefetch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A (no fwd prefetch):', d['ms_per_step'])" | tee -a $O/ab.txt; done
sed -i 's/#define CB_FWD_EPF 0/#define CB_FWD_EPF 1/' clipbert_amd/csrc/gemm_impl.h
python -m clipbert_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B (fwd residual prefetch):', d['ms_per_step'])" | tee -a $O/ab.txt; done
python -m pytest tests/test_kernels_gemm.py -q -m gpu -x 2>&1 | tail -2
