cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_reconstruct.py tests/test_gpu_texture.py tests/test_gpu_net.py -x -q 2>&1 | tail -3
O=gpurun_out/r07a_carry_ab.txt
echo "# bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-alt; RN_NO_CARRY=1 = the skip gradient through autograd's add_ (before), unset = added in conv1's input-gradient launch" > $O
for r in 1 2 3; do for c in 1 0; do
  RN_NO_CARRY=$c python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); p=d.get('parity') or {}; print('no_carry $c  %8.2f %s %8.3f ms/step  %s'%(d['value'],d['unit'],d['ms_per_step'],str({k:p[k] for k in p if 'err' in k})[:160]))" >> $O
done; done
cat $O
