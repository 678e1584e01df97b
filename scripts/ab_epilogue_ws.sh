cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_reconstruct.py tests/test_gpu_texture.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); p=d.get('parity') or {}; print('train %8.2f %s %8.3f ms/step  %s'%(d['value'],d['unit'],d['ms_per_step'],str({k:p[k] for k in p if 'err' in k})[:200]))"; done
