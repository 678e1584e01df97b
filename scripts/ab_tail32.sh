cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "conv2d_transpose" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q 2>&1 | tail -2
for i in 1 2; do for nt in 1 0; do
  if [ $nt = 1 ]; then export RN_NO_TAIL_KERNEL=1; else unset RN_NO_TAIL_KERNEL; fi
  python bench.py --mode stress --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('stress no_tail_kernel=$nt  %8.2f %s %8.3f ms/step  parity %s'%(d['value'],d['unit'],d['ms_per_step'],(d.get('parity') or {}).get('max_abs_err')))"
done; done
