#!/bin/bash
# PMC passes (SQ set | FETCH_SIZE | WRITE_SIZE | TCC hit/miss) over scripts/bf3_check.py's timing loop of the res2 shape:
# the split (bf16x3) input transform + GEMM stage next to the exact-fp32 ones.  Run on the GPU box.
#   usage: gpurun -- "bash scripts/pmc_bf3.sh"   ->  gpurun_out/pmc_bf3/bf3.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${OUT:-$R/gpurun_out/pmc_bf3}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
for tag in sq fetch write tcc; do
    case $tag in
        sq) C=$SQ ;; fetch) C="FETCH_SIZE" ;; write) C="WRITE_SIZE" ;; tcc) C=$TCC ;;
    esac
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/bf3.$tag" -o "$tag" -- python "$R/scripts/bf3_check.py" --no-accuracy --iters 3 ${BF3_ARGS:-} > "$OUT/bf3.$tag.log" 2>&1
done
: > "$OUT/bf3.txt"
for tag in sq fetch write tcc; do
    f=$(find "$OUT/bf3.$tag" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python "$R/scripts/pmc_summary.py" "$f" wino >> "$OUT/bf3.txt"
done
echo "wrote $OUT/bf3.txt"
