# same-box A/B of the split GEMM stage's launch plan: base = scripts/_build/librendernet_hip_base.so (the committed planner), default = the working tree
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/${1:-r06w}_ab_libs.txt
for args in "--batch 3" "--batch 4" "--batch 6" "--batch 24" "--mode train"; do
  APPEND=1 BENCH_ARGS="$args" bash scripts/ab_libs.sh ${1:-r06w} 2 base default
done
cat gpurun_out/${1:-r06w}_ab_libs.txt
