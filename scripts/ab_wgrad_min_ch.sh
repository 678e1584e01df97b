cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z_wgrad_min_ch.txt
echo "# bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-alt; RN_WGRAD_SPLIT_MIN_CH = 1024 (default) / 512" > $O
for r in 1 2 3; do for m in 1024 512; do
  RN_WGRAD_SPLIT_MIN_CH=$m python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('min_ch $m  %8.2f %s %8.3f ms/step  check %s'%(d['value'],d['unit'],d['ms_per_step'],str(d.get('grad_check') or d.get('parity'))[:120]))" >> $O
done; done
cat $O
