# same-box A/B of variant libraries (scripts/build_variant.py): usage  [BENCH_ARGS="--batch 3"] bash scripts/ab_libs.sh <tag> <rounds> <name>...   ("default" = the product build)
cd $GRAFT_REPO_ROOT
tag=$1; rounds=$2; shift 2
O=gpurun_out/${tag}_ab_libs.txt
[ -n "${APPEND:-}" ] || : > $O
for r in $(seq 1 $rounds); do
 for n in "$@"; do
  if [ $n = default ]; then unset RN_HIP_LIBRARY; else export RN_HIP_LIBRARY=$GRAFT_REPO_ROOT/scripts/_build/librendernet_hip_$n.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('%-8s %-16s %8.2f %s %8.3f ms/step  parity %s'%('$n','${BENCH_ARGS:-}',d['value'],d['unit'],d['ms_per_step'],(d.get('parity') or {}).get('max_abs_err')))" >> $O
 done
done
unset RN_HIP_LIBRARY
