cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y_w4_p16_ab.txt
echo "# RN_WINO_BF3_W4=1: whole items on the four-wave kernel (128x128 wave tiles), both in the 16x16x32 paired-piece form; bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt" > $O
for r in 1 2 3; do for w in 0 1; do
  RN_WINO_BF3_W4=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('W4=$w  %8.2f frames/s %8.3f ms/step  frac %.4f  parity %s'%(d['value'],d['ms_per_step'],d['roofline']['frac'],(d.get('parity') or {}).get('max_abs_err')))" >> $O
done; done
cat $O
RN_WINO_BF3_W4=1 timeout 900 python -m pytest tests/test_gpu_wino_split.py -x -q -k "not four_wave" 2>&1 | tail -2
