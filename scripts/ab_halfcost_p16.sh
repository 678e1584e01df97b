cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v_halfcost_p16.txt
echo "# 16x16x32 form: bench.py --batch B --steps 10 --warmup 3 --no-cpu-baseline --no-alt with RN_WINO_BF3_HALF_COST = 9 (default) / 11 / 13" > $O
for b in 3 2 4 5 6 3; do
 for hc in 9 11 13; do
  RN_WINO_BF3_HALF_COST=$hc python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('batch $b half_cost $hc  %8.2f frames/s %8.3f ms/step'%(d['value'],d['ms_per_step']))" >> $O
 done
done
cat $O
