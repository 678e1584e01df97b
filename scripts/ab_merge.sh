# A/B of the split GEMM stage's plan for the ragged last row block (profiles/r06n_gemm_merge_ab.txt): RN_WINO_BF3_NOMERGE=1 = always a launch of its own
# (rounds 5-6), default = one more block of the main launch when the cost model says so.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06n}_gemm_merge_ab.txt
echo "# bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt, one MI355X; nomerge 1 = the old plan" > $O
for b in 24 12 6 24 12 6; do
 for nm in 1 0; do
  if [ $nm = 1 ]; then export RN_WINO_BF3_NOMERGE=1; else unset RN_WINO_BF3_NOMERGE; fi
  python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('render  batch $b nomerge $nm  %8.2f frames/s %8.3f ms/step  parity %s'%(d['value'],d['ms_per_step'],(d.get('parity') or {}).get('max_abs_err')))" >> $O
 done
done
for m in texture train texture train; do
 for nm in 1 0; do
  if [ $nm = 1 ]; then export RN_WINO_BF3_NOMERGE=1; else unset RN_WINO_BF3_NOMERGE; fi
  python bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$m nomerge $nm  %8.2f %s %8.3f ms/step'%(d['value'],d['unit'],d['ms_per_step']))" >> $O
 done
done
unset RN_WINO_BF3_NOMERGE
cat $O
timeout 1500 python -m pytest tests/test_gpu_wino_split.py tests/test_gpu_net.py tests/test_gpu_train.py -x -q 2>&1 | tail -3
