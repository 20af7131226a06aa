cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py --drop-features "$1" --streams 1 --steps 20 --warmup 3 --cpu-sample 0 --latency-requests 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('drop=$1', 'cols', d['config']['columns'], {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()})
"; }
run profile
run divers_genres,divers_actors,divers_tags,divers_year,divers_popularity
run profile,divers_genres,divers_actors,divers_tags,divers_year,divers_popularity
run ctr,ctr_tag,ctr_genre
run popularity,vote_avg,vote_cnt,budget,release_date,runtime,title_length,genre
