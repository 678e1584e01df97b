cd $GRAFT_REPO_ROOT
export GIT_REV=$1
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r07e_suite.txt 2>&1; tail -2 gpurun_out/r07e_suite.txt
bash scripts/round_measure.sh r07e > gpurun_out/r07e_measure.log 2>&1; tail -12 gpurun_out/r07e_measure.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/r07e_pmc bash scripts/collect_pmc.sh > gpurun_out/r07e_pmc.log 2>&1; tail -1 gpurun_out/r07e_pmc.log
