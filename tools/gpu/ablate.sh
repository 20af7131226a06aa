#!/bin/bash
# ablation of the assembly kernel by feature family (bench.py --drop-features), one line per variant
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
run() { tag=$1; shift; timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --latency-requests 0 --streams 1 "$@" > gpurun_out/ablate/$tag.json 2> gpurun_out/ablate/$tag.log || tail -5 gpurun_out/ablate/$tag.log
python - <<PY
import json
d=json.load(open("gpurun_out/ablate/$tag.json"))
print("$tag".ljust(14), round(d['value']/1e6,1),'M items/s', round(d['ms_per_step'],3),'ms', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in d['kernels'].items()}, d['config']['columns'], d['config']['tile_columns'])
PY
}
run base
run no_div_num --drop-features divers_year,divers_popularity
run no_div_str --drop-features divers_genres,divers_actors,divers_tags
run no_div --drop-features divers_genres,divers_actors,divers_tags,divers_year,divers_popularity
run no_profile --drop-features profile
run no_rates --drop-features ctr,ctr_tag,ctr_genre
run no_cross --drop-features profile,divers_genres,divers_actors,divers_tags,divers_year,divers_popularity
run only_numbers --drop-features profile,divers_genres,divers_actors,divers_tags,divers_year,divers_popularity,ctr,ctr_tag,ctr_genre,genre,title_length,position
