#!/bin/bash
# Regenerate the counter evidence of the three roofline kernels on the CURRENT build (run on the GPU box):
#   * the dominant conv  -- the three launches of the Winograd F(6x6,3x3) path on the res2 shape (scripts/wino_bench.py
#     --shapes 64x1024 --only f63) and of F(4x4,3x3) on the same shape (--only f43; RN_NO_WINOGRAD63=1 runs the net on it)
#   * the same shape with the split (bf16x3) input transform and GEMM stage (scripts/bf3_check.py --shapes 0)
#   * the 3-D encoder layer (scripts/layer_bench.py --only res1): the direct depth-run kernel (RN_NO_WINOGRAD3D=1) and
#     the Winograd kernel on the same layer
#   * the resampler's three launches (scripts/layer_bench.py --only resample)
# One rocprofv3 --pmc pass per counter set (SQ set | FETCH_SIZE | WRITE_SIZE | TCC hit/miss), --kernel-trace only (no
# other trace domain beside --pmc).  Summaries land in $OUT (default gpurun_out/pmc); scripts/pmc_to_traffic.py then
# rewrites profiles/traffic.json with the csrc digest + git revision the numbers belong to.
#   usage (from the build container):  gpurun -- "GIT_REV=$(git rev-parse --short HEAD) bash scripts/collect_pmc.sh"
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${OUT:-$R/gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
run() {  # name, counter-set tag, counters..., then "--" and the command
    local name=$1 tag=$2; shift 2
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    rocprofv3 --kernel-trace --output-format csv --pmc "${ctrs[@]}" -d "$OUT/$name.$tag" -o "$tag" -- "$@" > "$OUT/$name.$tag.log" 2>&1
}
for tag in sq fetch write tcc; do
    case $tag in
        sq) C=$SQ ;; fetch) C="FETCH_SIZE" ;; write) C="WRITE_SIZE" ;; tcc) C=$TCC ;;
    esac
    # PMC_SET=full: every kernel family of rounds 2-4; default (round 5): the kernels the bench lines quote `traffic` for (the res2 GEMM stage in
    # all three modes: bf3_check.py runs the fp32 stages, then the split ones), the 3-D split kernel and the resampler
    if [ "${PMC_SET:-}" = full ]; then
    run wino63 $tag $C -- python "$R/scripts/wino_bench.py" --shapes 64x1024 --iters 3 --only f63
    run wino43 $tag $C -- python "$R/scripts/wino_bench.py" --shapes 64x1024 --iters 3 --only f43
    RN_NO_WINOGRAD3D=1 run res1 $tag $C -- python "$R/scripts/layer_bench.py" --only res1 --iters 3
    RN_WINO_GEMM=f32 run res1w $tag $C -- python "$R/scripts/layer_bench.py" --only res1 --iters 3
    fi
    run bf3 $tag $C -- python "$R/scripts/bf3_check.py" --no-accuracy --iters 3 --shapes 0          # res2 shape, F(6x6,3x3): fp32 stages, then the split ones
    RN_CONV3D_SPLIT=1 run res1s $tag $C -- python "$R/scripts/layer_bench.py" --only res1 --iters 3        # the bf16x3 kernel on the same layer
    run resample $tag $C -- python "$R/scripts/layer_bench.py" --only resample --iters 5 --no-dense
done
for name in wino63 wino43 bf3 res1 res1w res1s resample; do
    flt=""; [ $name = wino63 ] && flt=wino; [ $name = wino43 ] && flt=wino; [ $name = bf3 ] && flt=wino_; [ $name = res1 ] && flt=conv3d_k3; [ $name = res1w ] && flt=conv_wino; [ $name = res1s ] && flt=conv3d_wino_bf3; [ $name = resample ] && flt=resample_
    : > "$OUT/$name.txt"
    for tag in sq fetch write tcc; do
        f=$(find "$OUT/$name.$tag" -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && python "$R/scripts/pmc_summary.py" "$f" "$flt" >> "$OUT/$name.txt"
    done
done
python "$R/scripts/pmc_to_traffic.py" "$OUT" "$OUT/traffic.json"
echo "wrote $OUT/{wino63,wino43,bf3,res1,res1w,res1s,resample}.txt and $OUT/traffic.json"
