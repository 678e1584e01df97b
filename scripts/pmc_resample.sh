cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d /tmp/pmc_rs -o sq -- python $R/scripts/resample_bench.py --iters 5 > /tmp/pmc_rs.log 2>&1
f=$(find /tmp/pmc_rs -name "*counter_collection.csv" | head -1)
python $R/scripts/pmc_summary.py $f resample_ > $R/gpurun_out/r06b_pmc_rs.txt
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_rs2 -o sq -- python $R/scripts/resample_bench.py --iters 5 > /tmp/pmc_rs2.log 2>&1
f=$(find /tmp/pmc_rs2 -name "*counter_collection.csv" | head -1)
python $R/scripts/pmc_summary.py $f resample_ >> $R/gpurun_out/r06b_pmc_rs.txt
head -60 $R/gpurun_out/r06b_pmc_rs.txt
