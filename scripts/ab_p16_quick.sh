cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06p3}_p16_quick.txt
: > $O
for p in 0 1 0 1; do
  echo "## RN_WINO_BF3_P16=$p" >> $O
  RN_WINO_BF3_P16=$p python scripts/bf3_check.py --no-accuracy --batch 24 --shapes 0,1,2 2>&1 | grep "split " | grep -v split16 >> $O
  RN_WINO_BF3_P16=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('bench P16=$p  %8.2f frames/s %8.3f ms/step  frac %.4f  parity %s'%(d['value'],d['ms_per_step'],d['roofline']['frac'],(d.get('parity') or {}).get('max_abs_err')))" >> $O
done
cat $O
