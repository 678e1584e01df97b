cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --hip-runtime-trace --memory-copy-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-alt > /tmp/tr.log 2>&1
ls /tmp/tr
python - <<'PY'
import csv, collections, glob
for f in glob.glob('/tmp/tr/*hip_api_trace.csv'):
    c=collections.Counter(r['Function'] for r in csv.DictReader(open(f)))
    for k,v in c.most_common(25): print(v,k)
for f in glob.glob('/tmp/tr/*memory_copy_trace.csv'):
    rows=list(csv.DictReader(open(f)))
    print(len(rows), rows[0].keys() if rows else '')
    c=collections.Counter((r.get('Direction'), r.get('Bytes') or r.get('Size')) for r in rows)
    for k,v in c.most_common(15): print(v,k)
PY
