# A/B of the split GEMM stage's MFMA shape (profiles/r06p_*): RN_WINO_BF3_P16=1 = v_mfma_f32_16x16x32_bf16 with paired pieces, 0 = v_mfma_f32_32x32x16_bf16
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06p}_p16_ab.txt
: > $O
for p in 0 1 0 1; do
  echo "## RN_WINO_BF3_P16=$p  scripts/bf3_check.py --no-accuracy --batch 24" >> $O
  RN_WINO_BF3_P16=$p python scripts/bf3_check.py --no-accuracy --batch 24 2>&1 | grep -v amdgpu.ids | grep "split \|shape\|^#\|x" | grep -v split16 >> $O
done
echo "## accuracy: RN_WINO_BF3_P16=1 scripts/bf3_check.py --batch 24 --shapes 0" >> $O
RN_WINO_BF3_P16=1 python scripts/bf3_check.py --batch 24 --shapes 0 2>&1 | grep -v amdgpu.ids >> $O
echo "## accuracy: RN_WINO_BF3_P16=0" >> $O
RN_WINO_BF3_P16=0 python scripts/bf3_check.py --batch 24 --shapes 0 2>&1 | grep -v amdgpu.ids >> $O
for p in 0 1 0 1; do
  RN_WINO_BF3_P16=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('bench P16=$p  %8.2f frames/s %8.3f ms/step  frac %.4f  parity %s'%(d['value'],d['ms_per_step'],d['roofline']['frac'],(d.get('parity') or {}).get('max_abs_err')))" >> $O
done
cat $O
RN_WINO_BF3_P16=1 timeout 1500 python -m pytest tests/test_gpu_wino_split.py tests/test_gpu_net.py -x -q 2>&1 | tail -3
