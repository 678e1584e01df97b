# A/B of the split GEMM stage's item plan at small batches (profiles/r06l_halfcost.txt): RN_WINO_BF3_HALF_COST=8 reproduces the plan of rounds 5-6
# (a round of half items = half a round of whole items, launches free); the default prices a half round at 9/16 and a further launch at 4/16.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06l}_halfcost.txt
echo "# bench.py --batch B --steps 10 --warmup 3 --no-cpu-baseline --no-alt, one MI355X; half_cost 8 = the old plan, 9 = the default" > $O
for b in 2 3 4 5 2 3; do
 for hc in 8 9; do
  RN_WINO_BF3_HALF_COST=$hc python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('batch $b half_cost $hc  %8.2f frames/s %8.3f ms/step'%(d['value'],d['ms_per_step']))" >> $O
 done
done
cat $O
