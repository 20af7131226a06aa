cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MRK_BENCH_FORCE_DIST=1 MASTER_PORT=29531 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2 dist-leg', round(d['value']/1e6,1), d['ms_per_step'], d['config']['parallelism'])"
MRK_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --workload c4 --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 dist-leg', round(d['value']/1e6,1), d['ms_per_step'], d['config']['parallelism'])"
