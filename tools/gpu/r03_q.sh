# round 3, pass q: what limits the end-to-end loop (fresh requests per batch)? host threads x batches in flight
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_q
mkdir -p $O
for t in 1 2; do for nb in 3 5; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --e2e-seconds 1.5 --e2e-threads $t --e2e-batches $nb > $O/b_t${t}_n$nb.json 2> $O/b_t${t}_n$nb.log || tail -3 $O/b_t${t}_n$nb.log
  python - $t $nb $O/b_t${t}_n$nb.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    e = d['e2e']
    print('threads', sys.argv[1], 'in flight', sys.argv[2], 'value', round(d['value']/1e6, 1), 'e2e', round(e['value']/1e6, 1), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in e.items() if k not in ('value',)})
except Exception as ex:
    print(sys.argv[1], sys.argv[2], 'FAILED', ex)
PY
done; done
