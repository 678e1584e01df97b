#!/bin/bash
# Everything the round's evidence under profiles/ comes from, in one pass on the GPU box (copy gpurun_out/$TAG_* to profiles/):
#   bench lines of the four modes (un-profiled), rocprofv3 per-shape tables of render / texture / train, the batch sweep,
#   the stage A/B and 3-D layer timings, the split (bf16x3) route's per-shape tables.   usage: gpurun -- "bash scripts/round_measure.sh r03c"
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
python bench.py --steps 20 --warmup 3 > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"
python bench.py --mode texture --steps 20 --warmup 3 > "$O/${TAG}_bench_texture.json" 2>> "$O/${TAG}_bench.err"
python bench.py --mode stress --steps 5 --warmup 2 > "$O/${TAG}_bench_stress.json" 2>> "$O/${TAG}_bench.err"
python bench.py --mode train --steps 10 --warmup 3 > "$O/${TAG}_bench_train.json" 2>> "$O/${TAG}_bench.err"
{
  echo "# python bench.py --batch B --steps 10 --warmup 3 --no-cpu-baseline, one MI355X, git ${GIT_REV:-?}; value = the default mode (bf16x3 split multiply stages)"
  for b in 1 3 6 12 24 48; do
    python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
e = d.get('exact') or {}
a2 = d.get('alt2') or {}
print('batch %3d  default (bf16x3) %8.2f frames/s  %8.2f ms/step  frac %.3f (of bf16 peak / 6)   |   exact fp32 %8.2f frames/s  %8.2f ms/step  frac %.3f (of the fp32 MFMA peak)   |   split16 (fp16x2) %8.2f frames/s  %8.2f ms/step  frac %.3f (of fp16 peak / 3)'
      % ($b, d['value'], d['ms_per_step'], d['roofline']['frac'], e.get('value', 0), e.get('ms_per_step', 0), (e.get('roofline') or {}).get('frac', 0),
         a2.get('value', 0), a2.get('ms_per_step', 0), (a2.get('roofline') or {}).get('frac', 0)))"
  done
} > "$O/${TAG}_batch_sweep.txt"
{
  echo "# scripts/bf3_check.py --no-accuracy: the three stages, exact-fp32 multiply stage vs bf16x3 / fp16x2 split; one MI355X, git ${GIT_REV:-?}"
  python scripts/bf3_check.py --no-accuracy --batch 24 2>&1 | grep -v amdgpu.ids
  echo "# scripts/wgrad_split_bench.py: F(4x4,3x3) / F(4x4,4x4) filter gradient at crop 64, exact fp32 vs split (all four launches)"
  python scripts/wgrad_split_bench.py 2>&1 | grep -v amdgpu.ids
  echo "# scripts/c3_check.py: fused 3x3x3 32 -> 32 kernel, fp32 vs bf16x3 / fp16x2 split"
  python scripts/c3_check.py --no-accuracy 2>&1 | grep -v amdgpu.ids
  echo "# scripts/latency_bench.py: single-frame latency, eager vs hipGraph replay, default mode"
  python scripts/latency_bench.py 2>&1 | grep -v amdgpu.ids
} > "$O/${TAG}_stage_ab.txt"
bash scripts/profile_bench.sh ${TAG}_render --steps 5 --warmup 2 --no-cpu-baseline --no-alt > /dev/null 2>&1                       # the default mode (bf16x3)
bash scripts/profile_bench.sh ${TAG}_render_exact --gemm f32 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > /dev/null 2>&1
bash scripts/profile_bench.sh ${TAG}_render_split16 --gemm split16 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > /dev/null 2>&1
bash scripts/profile_bench.sh ${TAG}_texture --mode texture --steps 5 --warmup 2 --no-cpu-baseline --no-alt > /dev/null 2>&1
bash scripts/profile_bench.sh ${TAG}_train --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-alt > /dev/null 2>&1
bash scripts/profile_bench.sh ${TAG}_b3 --batch 3 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > /dev/null 2>&1
tail -3 "$O/${TAG}_bench.err"
cut -c1-250 "$O/${TAG}_bench.json"
cat "$O/${TAG}_batch_sweep.txt"
