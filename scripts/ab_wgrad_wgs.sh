cd $GRAFT_REPO_ROOT
for w in 3072 1536 768 3072 6144; do
  RN_WGRAD_WGS=$w python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('RN_WGRAD_WGS=$w  %8.2f %s %8.3f ms/step  wgrad3d %s'%(d['value'],d['unit'],d['ms_per_step'],str((d.get('roofline_wgrad') or {}).get('avg_launch_ms'))))"
done
