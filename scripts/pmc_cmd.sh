#!/bin/bash
# rocprofv3 --pmc (SQ set, then LDS / instruction-mix set) of any command, summarised with scripts/pmc_summary.py.
#   usage (GPU box): bash scripts/pmc_cmd.sh <tag> <kernel-name filter> <command...>   ->  gpurun_out/<tag>_pmc.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; flt=$2; shift 2
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
: > "$R/gpurun_out/${tag}_pmc.txt"
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_${tag}_$i
  ( cd "$R" && rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc_${tag}_$i -o c -- "$@" ) > /tmp/pmc_${tag}_$i.log 2>&1
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/scripts/pmc_summary.py" "$f" "$flt" >> "$R/gpurun_out/${tag}_pmc.txt"
done
cat "$R/gpurun_out/${tag}_pmc.txt"
