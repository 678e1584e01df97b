cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/scripts/recon_step_bench.py 2>&1 | tail -1
rm -rf /tmp/prof_recon
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_recon -o p -- python $GRAFT_REPO_ROOT/scripts/recon_step_bench.py > /tmp/recon.log 2>&1
tr=$(find /tmp/prof_recon -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/summarize_trace.py "$tr" > $GRAFT_REPO_ROOT/gpurun_out/r07i_recon_per_shape.md
head -34 $GRAFT_REPO_ROOT/gpurun_out/r07i_recon_per_shape.md | cut -c1-175
