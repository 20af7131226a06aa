#!/usr/bin/env python
"""Benchmark of the /rank hot path (feature assembly + LambdaMART scoring + ordering) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over `--batches-per-step` device batches of synthetic Ranklens-shaped requests
that are already resident in HBM (resolved, uploaded): per device batch, assembly (pre-pass + gather, straight into the
scorer's binned tile) -> score -> sort for `--requests` requests of `--items` candidate items each.  The default
(c2: 96 device batches = 368 640 requests = 36.9 M candidates per step) makes the driver's 20-step run a ~1 s timed
region instead of 11 ms; the resident requests are `--streams` distinct device batches visited in turn, each on its own
HIP stream.  `value` is that device-resident throughput (the contract: inputs in HBM when the clock starts).

The JSON line also carries "e2e": the serving loop with FRESH requests every device batch - mrk_batch_load (host part
of the request + upload of the id bytes; item ids are resolved to store slots by a kernel) -> run -> download of
scores / order / status into pinned memory, `--e2e-threads` host threads (default 2) with `--e2e-batches` batches in
flight each, >= 1 s of timed work.

--workload c2 (default)  the configuration BASELINE.json's metric is quoted on: 100-item requests, the
             24 Ranklens columns (stock Ranklens model), 500-tree LightGBM-format LambdaMART;
             3840 x 100 = 384 000 items per GPU per step (3000 scorer wavefronts of 128 items: one
             resident round on 256 CUs x 12 waves).  Weak scaling: every rank owns a replica of the
             feature store and its own requests; for N > 1 the per-step scores are merged with one RCCL
             all-gather (the only exchange the path has).
--workload c3   1000-item requests, 64 mixed columns (BASELINE config 3), 384 requests per GPU per step.
--workload c5   c2 plus one bi-encoder column (BASELINE config 5): every request carries its own query TEXT; a step =
                the device forward pass of the step's queries (all-MiniLM-L6-v2's architecture, random weights, fp16)
                followed by the rank batch.  Its JSON carries an extra "encoder" object.
--workload c4   ONE request with 100 000 candidates (BASELINE config 4), item-sharded over the ranks:
             every rank assembles + scores its slice, one in-place RCCL all-gather of the f64 scores,
             then the sort.  Strong scaling ("scaling": "strong").
--workload c4x  the out-of-cache form of c4: the generated catalogue is cloned to 8 M items (3 GB of item records, no
             cache level holds it) and ONE request carries 4 M candidates drawn from all of them - the gather of the
             assembly kernel against real HBM.  No CPU baseline / latency leg (the oracle does not hold 8 M items).

One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
  roofline     dominant kernel, algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline the CPU oracle (a scalar port of the reference's read path) on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}  # dense matrix peaks by OPERAND type (f32-input MFMA = the f32 vector rate)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c4x", "c5"], default="c2")
    ap.add_argument("--encoder-precision", choices=["f32", "f16"], default="f32",
                    help="c5: arithmetic of the text encoder in the timed region (f32 = mrk_encoder_load's default, the fp32 ONNX session's "
                         "arithmetic on the f32-input matrix instruction; f16 = BASELINE config 5's opt-in); the other one is timed beside it")
    ap.add_argument("--clones", type=int, default=79,
                    help="c4x: deep copies of every generated item (mrk_debug_clone_items): 100 000 x (1 + 79) = 8 M items, a 3 GB item "
                         "table - far beyond the 256 MB Infinity Cache")
    ap.add_argument("--requests", type=int, default=None, help="requests per step per GPU (c2: 3840, c3: 384, c4: 1)")
    ap.add_argument("--items", type=int, default=None, help="candidate items per request (c2: 100, c3: 1000, c4: 100000)")
    ap.add_argument("--streams", type=int, default=None,
                    help="batches in flight per GPU (each on its own HIP stream; steps alternate between them). "
                         "Default 3 for c2 / c3 (the assembly kernels wait on memory while the scorer is VALU-bound: consecutive "
                         "batches overlap; round 6, same box: 2 / 3 / 4 in flight = 1 168 / 1 191 / 1 198 M items/s on c2, 875 / 952 M on c3 - "
                         "profiles/r06_t_streams.txt; round 5's kernels lost with 3), 1 for c4")
    ap.add_argument("--batches-per-step", type=int, default=None,
                    help="device batches one step runs (default: c2 / c3 96, c4 128, c5 8: a step is about 50 ms of device work)")
    ap.add_argument("--e2e-seconds", type=float, default=1.5, help="timed length of the end-to-end serving loop (0 = skip)")
    ap.add_argument("--e2e-batches", type=int, default=4, help="batches in flight per host thread in the end-to-end loop")
    ap.add_argument("--e2e-threads", type=int, default=2,
                    help="host threads driving the end-to-end loop (each its own batches).  Round 4: with the device batch at 0.335 ms the "
                         "0.28 ms of host work per batch (mrk_batch_load) make ONE thread the limit of the loop - 1 028 M items/s = 0.90 of the "
                         "device-resident rate, two threads 1 116 M = 0.97 (same box, profiles/r04_j_bench_c2_e2e2.json); while a batch took "
                         "0.41 ms one thread kept up and more only added GIL hand-overs (round 3: 683 / 592 / 636 M with 1 / 2 / 3)")
    ap.add_argument("--e2e-sets", type=int, default=6, help="distinct request sets the end-to-end loop cycles through")
    ap.add_argument("--drop-features", default="", help="experiments only: comma-separated features removed from the model")
    ap.add_argument("--catalogue", type=int, default=100_000)
    ap.add_argument("--sessions", type=int, default=10_000)
    ap.add_argument("--trees", type=int, default=None, help="default 500 (lightgbm) / 100 (xgboost)")
    ap.add_argument("--backend", choices=["lightgbm", "xgboost"], default="lightgbm",
                    help="booster format of the synthetic forest.  lightgbm (default): 500 leaf-wise 16-leaf trees, f64 - the Ranklens "
                         "stock model's backend.  xgboost: complete depth-`--depth` trees, f32 - BASELINE config #2 as written is "
                         "`--backend xgboost --trees 100 --depth 6`")
    ap.add_argument("--quantiles", type=int, default=49,
                    help="split candidates per column of the synthetic forest (empirical quantiles of a sample matrix).  49 (default, rounds 1-6): "
                         "the forest ends up with ~50 distinct thresholds on a continuous column - 6 KB of threshold tables, which the assembly "
                         "kernels keep resident in LDS (round 6); 254 = a LightGBM model trained with max_bin 255 whose 7 500 splits use most "
                         "bins: ~40 KB of tables - the workgroup-per-request kernels stage them per column, the item-parallel kernel still holds them")
    ap.add_argument("--depth", type=int, default=6, help="xgboost: tree depth (Metarank's XGBoost default maxDepth is 8)")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="requests timed on the CPU oracle (0 = skip)")
    ap.add_argument("--latency-requests", type=int, default=300, help="single-request latency samples (0 = skip, also skips the sweep)")
    ap.add_argument("--concurrent-callers", default=None, help="caller counts of the concurrent-callers leg of `latency` ('' = skip)")
    ap.add_argument("--concurrent-requests", type=int, default=400, help="... requests per caller")
    ap.add_argument("--latency-sweep", type=int, default=None,
                    help="sequential requests per size of the reference's latency protocol (sizes 25..300; default 5000 for the default c2 run, 0 = skip)")
    args = ap.parse_args()
    wl = args.workload
    if args.requests is None:
        args.requests = {"c2": 3840, "c3": 384, "c4": 1, "c4x": 1, "c5": 3840}[wl]
    if args.items is None:
        args.items = {"c2": 100, "c3": 1000, "c4": 100_000, "c4x": 4_000_000, "c5": 100}[wl]
    sharded = wl in ("c4", "c4x")
    if args.latency_sweep is None:
        args.latency_sweep = 5000 if (wl == "c2" and args.gpus == 1) else 0
    if args.concurrent_callers is None:
        args.concurrent_callers = "1,16,64,128,256" if (wl == "c2" and args.gpus == 1) else ""
    if wl == "c4x":
        args.cpu_sample, args.latency_requests = 0, 0
    if args.trees is None:
        args.trees = 500 if args.backend == "lightgbm" else 100
    if args.streams is None:
        args.streams = 1 if sharded else 3
    n_streams = max(1, args.streams)
    if args.batches_per_step is None:
        args.batches_per_step = {"c2": 96, "c3": 96, "c4": 128, "c4x": 4, "c5": 8}[wl]
    bps = max(1, args.batches_per_step)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log(f"WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = max(world, 1)

    # multi-GPU: one process per GPU (the driver's torchrun launch provides RANK / WORLD_SIZE / MASTER_*), the RCCL
    # communicator lives inside the library (csrc/comm.cpp) - no torch anywhere.  MRK_BENCH_FORCE_DIST=1 exercises the
    # collective leg on one GPU (a world of one goes through RCCL all the same).
    use_dist = n_gpus > 1 or bool(os.environ.get("MRK_BENCH_FORCE_DIST"))

    import metarank_amd as M
    from workloads import ranklens, synth

    if os.environ.get("MRK_BENCH_DRY_DIST"):
        # No device: walk the multi-process start-up as far as it goes without one (tests/test_dist_cpu.py launches this
        # with 2 ranks on the CPU box) - the communicator id hand-over between the launcher's children, the argument checks of
        # mrk_comm_init, and the library's own shard arithmetic for this workload's request.
        import ctypes as C
        from metarank_amd import _native as N
        from metarank_amd.dist import exchange_unique_id_file

        L = N.lib()

        def make_id():
            buf = (C.c_uint8 * 128)()
            if L.mrk_comm_unique_id(buf) != N.MRK_OK:   # RCCL may refuse to draw an id on a host without a GPU
                return bytes((7 * i + 1) & 0xff for i in range(128))
            return bytes(buf)

        comm_key = f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_dry"
        uid = exchange_unique_id_file(rank, n_gpus, make_id, comm_key)
        assert len(uid) == 128
        rc = L.mrk_comm_init(None, uid, rank, n_gpus)   # no context: the call must get as far as its argument checks
        assert rc == N.ERR_INVALID_ARG, rc
        items = args.items or {"c2": 100, "c3": 1000, "c4": 100_000, "c4x": 100_000 * (args.clones + 1) // 2, "c5": 100}[wl]
        lo, hi = C.c_int64(), C.c_int64()
        assert L.mrk_shard_range(items, rank, n_gpus, C.byref(lo), C.byref(hi)) == N.MRK_OK
        print(json.dumps({"dry": True, "rank": rank, "world": n_gpus, "uid_crc": sum(uid) & 0xffff, "items": items,
                          "chunk": int(L.mrk_shard_chunk(items, n_gpus)), "lo": lo.value, "hi": hi.value}), flush=True)
        return

    ctx = M.Context(local_rank)
    if use_dist:
        from metarank_amd.dist import exchange_unique_id_file

        # rank 0's ncclUniqueId to the other ranks of this node (all are children of the same launcher process)
        comm_key = f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
        # (RCCL prints a version banner on stdout when the first communicator comes up: stdout carries the ONE JSON line
        # of the contract, so file descriptor 1 points at stderr while the communicator is built)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = exchange_unique_id_file(rank, n_gpus, M.Context.comm_unique_id, comm_key)
            ctx.comm_init(uid, rank, n_gpus)
            ctx.comm_barrier()
        finally:
            sys.stdout.flush()
            try:   # RCCL printed through C stdio: its buffer must be emptied while descriptor 1 still points at stderr
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    cfg = ranklens.c3_config() if wl == "c3" else ranklens.c5_config() if wl == "c5" else ranklens.ranklens_config()
    if args.drop_features:
        drop = set(args.drop_features.split(","))
        cfg["features"] = [f for f in cfg["features"] if f["name"] not in drop]
        cfg["models"]["xgboost"]["features"] = [f for f in cfg["models"]["xgboost"]["features"] if f not in drop]
    ranker = M.HipRanker(cfg, ctx)
    model_name = "xgboost"
    dim = ranker.dim(model_name)
    enc = tok = None
    if wl == "c5":  # no network for checkpoints: the architecture of all-MiniLM-L6-v2 with random weights
        from metarank_amd.encoder import HipEncoder, HipTokenizer
        tok_json = synth.wordpiece_tokenizer_json(vocab_size=2000, max_length=128)
        # f32 (mrk_encoder_load's default): exact-f32 products on v_mfma_f32_16x16x4_f32, the same bits for a query alone and in a packed
        # batch - scores within 1e-5 / identical order hold; fp16 (BASELINE config 5's wording) is the opt-in and is timed beside it below
        enc_weights = synth.bert_safetensors(synth.synthetic_bert(), 12)
        enc = HipEncoder(enc_weights, tok_json, ctx=ctx, precision=args.encoder_precision)
        tok = HipTokenizer(tok_json)
        ranker.bind_encoder("title_match", enc)

    # ---- state: one generation pass feeds the device store (and the CPU oracle on rank 0, N=1)
    oracle = None
    do_cpu = rank == 0 and n_gpus == 1 and args.cpu_sample > 0
    if do_cpu:
        from oracle.assembly import OraclePlan, OracleStore, sort_order
        from oracle.forest import OracleForest

        class _O:
            pass

        oracle = _O()
        oracle.store = OracleStore()
        oracle.plan = OraclePlan(cfg, model_name)
    t0 = time.perf_counter()
    n_puts = 0
    fn_h = {"double": ranker.put_double, "string": ranker.put_string, "string_list": ranker.put_string_list,
            "double_list": ranker.put_double_list, "counter": ranker.put_counter, "periodic": ranker.put_periodic,
            "bounded_list": ranker.put_bounded_list}
    if oracle is not None:
        s = oracle.store
        fn_o = {"double": s.put_double, "string": s.put_string, "string_list": s.put_string_list,
                "double_list": s.put_double_list, "counter": s.put_counter, "periodic": s.put_periodic,
                "bounded_list": s.put_bounded_list}
    import itertools
    state = ranklens.generate_state(args.catalogue, args.sessions, c3=(wl == "c3"))
    if wl == "c5":
        state = itertools.chain(state, ranklens.c5_embeddings(args.catalogue))
    for kind, key, value in state:
        fn_h[kind](key, value)
        if oracle is not None:
            fn_o[kind](key, value)
        n_puts += 1
    n_catalogue = args.catalogue
    if wl == "c4x":
        n_catalogue = ranker.clone_items(args.clones)
    ranker.flush()
    log(f"state: {n_puts} feature values for {args.catalogue} items / {args.sessions} sessions in {time.perf_counter() - t0:.1f}s"
        + (f"; cloned to {n_catalogue} items" if wl == "c4x" else ""))

    # ---- requests (per-rank seeds) and the model
    # item-sharded: every rank holds the same request; else per-rank (and per-stream) requests
    if wl == "c4x":  # candidates drawn from the originals and all their clones
        def big_request(seed):
            rng = np.random.default_rng(seed)
            base = rng.integers(0, args.catalogue, args.items)
            k = rng.integers(0, args.clones + 1, args.items)
            ev = ranklens.generate_requests(1, 1, args.catalogue, args.sessions, seed=seed)[0]
            ev["items"] = [{"id": f"{b}#{c}" if c else f"{b}"} for b, c in zip(base.tolist(), k.tolist())]
            return ev
        all_events = [[big_request(ranklens.SEED + 1 + 1000 * k)] for k in range(n_streams)]
    else:
        all_events = [ranklens.generate_requests(args.requests, args.items, args.catalogue, args.sessions,
                                                 seed=ranklens.SEED + 1 + (0 if sharded else rank) + 1000 * k) for k in range(n_streams)]
    events = all_events[0]
    qtok = None
    if wl == "c5":  # a distinct query per request; the step's forward pass runs over these token ids
        qtok = []
        for k, evs in enumerate(all_events):
            texts = synth.synthetic_queries(len(evs), seed=100 + rank + 1000 * k)
            for ev, q in zip(evs, texts):
                ev["fields"] = [{"name": "query", "value": q}]
            qtok.append(tok.encode_batch(texts))
    sample_events = ranklens.generate_requests(64, 100, args.catalogue, args.sessions, seed=ranklens.SEED + 99)
    if wl == "c5":  # the cosine column's thresholds come from real cosines: without a query the column is NaN and the forest would never look at it
        for ev, q in zip(sample_events, synth.synthetic_queries(len(sample_events), seed=77)):
            ev["fields"] = [{"name": "query", "value": q}]
    sample = ranker.prepare(model_name, sample_events)
    sample.run(None)
    _, _, sm = sample.fetch(matrix=True)
    sample.close()
    if args.backend == "lightgbm":
        blob = synth.synthetic_lgbm_model(n_trees=args.trees, n_features=dim, num_leaves=16, max_depth=8,
                                          quantiles=ranklens.column_quantiles(sm, n=args.quantiles), cat_features=[7] if not args.drop_features else None, cat_prob=0.007,  # ~ one categorical split per 10 trees
                                          missing="per_feature")  # one missing type per column, as LightGBM's bin mappers produce
        booster = M.HipBooster(blob, M.LIGHTGBM, ctx)
    else:  # complete depth-d trees, f32 thresholds / leaves, one categorical split per ~10 trees
        blob = synth.synthetic_xgb_model(n_trees=args.trees, n_features=dim, depth=args.depth, quantiles=ranklens.column_quantiles(sm, n=args.quantiles),
                                         cat_features=[7] if not args.drop_features else None, cat_prob=0.1 / max(1, 2 ** args.depth - 1))
        booster = M.HipBooster(blob, M.XGBOOST, ctx)
    info = booster.info()
    t0 = time.perf_counter()
    if wl == "c4x":  # millions of ids: the flat form (no per-item C strings), resolved on the device
        batches = []
        for ev in all_events:
            bt = ranker.new_batch()
            bt.load(model_name, M.RequestSet(ev))
            bt.sync()
            batches.append(bt)
    else:
        batches = [ranker.prepare(model_name, ev) for ev in all_events]
    batch = batches[0]
    total_items = batch.total_items
    log(f"batch: {args.requests} requests x {args.items} items resolved + uploaded in {time.perf_counter() - t0:.2f}s; "
        f"model {info['n_trees']} trees, {info['n_nodes']} nodes, {info['device_bytes']} B on device")

    chunk = batch.shard_chunk(n_gpus) if sharded else total_items

    def sync_all():
        for bt in batches:
            bt.sync()
        ctx.sync()

    def barrier():
        if use_dist:
            ctx.comm_barrier()

    cur_enc = [enc]   # c5: the encoder whose forward pass a step runs (the other precision is timed after the main region)

    def run_one(i):
        bt = batches[i % n_streams]
        if qtok is not None:
            cur_enc[0].embed_ids(*qtok[i % n_streams])
        if sharded:   # this rank's slice -> one in-place RCCL all-gather of the f64 scores -> order (csrc/comm.cpp)
            bt.run_sharded(booster) if use_dist else bt.run(booster)
        else:
            bt.run(booster)
            if use_dist:   # replicas: the ranks' scores merged on every rank, inside the timed region
                bt.gather_scores()

    def step(i):
        for j in range(bps):
            run_one(i * bps + j)

    # Serve.maybeWarmup: the kernels specialised for this model come from the code objects shipped next to the library
    # (stock Ranklens program), else from the user's cache, else from a BACKGROUND compile (the library's default: no
    # request waits for the compiler) - run the batch shapes once, wait for those compiles, then warm up on them
    for j in range(n_streams):
        run_one(j)
    sync_all()
    ranker.warmup_kernels(model_name)
    for i in range(args.warmup):
        step(i)
    sync_all()
    barrier()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        elapsed = ctx.comm_max(elapsed)
    for bt in batches:
        st = bt.status()
        assert (st == 0).all(), f"requests failed: {st[st != 0][:5]}"
    ms_per_step = elapsed / args.steps * 1e3
    value = total_items * bps * (1 if sharded else n_gpus) * args.steps / elapsed
    ms_per_batch = ms_per_step / bps

    # ---- per-kernel HIP-event timing (outside the timed region, one batch at a time: the events add a little
    #      overhead and overlapping batches would stretch each other's kernels)
    ctx.profile_enable(True)
    prof_steps = max(5, min(args.steps, 20))
    for _ in range(prof_steps):
        if qtok is not None:   # c5: the forward pass of the batch's queries ("encoder": events on the encoder's own stream)
            enc.embed_ids(*qtok[0])
        if sharded:
            batch.run_shard(booster, rank, n_gpus)
            batch.sort()
        else:
            batch.run(booster)
        batch.sync()
    kernels = {}
    for k in ("encoder", "prepass", "assemble", "override", "bin", "score", "sort", "rank_fused"):
        ms, n = ctx.profile_get(k)
        if n:
            kernels[k] = {"avg_ms": ms / n, "launches_per_step": n / prof_steps}
    ctx.profile_enable(False)
    if os.environ.get("MRK_BENCH_PHASE"):   # measurement builds (MRK_DEFINES=MRK_PHASE_CLOCKS): where the specialised kernel's cycles go
        import ctypes as C
        lib_ = M.lib()
        lib_.mrk_debug_phase.restype = C.c_int
        lib_.mrk_debug_phase.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        out_ = (C.c_uint64 * 64)()
        rc_ = lib_.mrk_debug_phase(ctx._h, model_name.encode(), out_)
        v_ = [float(x) for x in out_]
        wg_ = max(v_[6], 1.0)
        log(f"phase clocks rc={rc_} workgroups={int(wg_)} per workgroup (thread 0): " +
            ", ".join(f"[{i}] {v_[i] / wg_:.0f}" for i in (0, 1, 2, 3, 4, 5, 7)))
        feats_ = cfg["models"][model_name]["features"]
        log("per op: " + ", ".join(f"{n} {v_[16 + i] / wg_:.0f}" for i, n in enumerate(feats_)))
    for k in kernels:
        kernels[k]["launches_per_batch"] = kernels[k].pop("launches_per_step")
    dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_batch"])
    # Algorithmic bytes per launch of the dominant kernel = SURVEY.md 8(d)'s per-item figure x the items one launch
    # processes (+ the model, read once per launch).  B_item for the configuration as built (f64 columns, LightGBM):
    #   store read  8 B x D matrix columns + ~48 B of list tokens (12 tokens x 4 B)
    #   ids / slot in 4 B, score out 8 B
    #   c5: + the item's stored embedding, 384 x 4 B (float-exact double lists are kept - and read - as f32)
    # No intermediate tile, no pre-pass tokens: what the fused path would move at best.
    V = info["tile_columns"]                  # u16 cells per item in the scorer's tile
    my_items = min(chunk, total_items) if sharded else total_items
    b_item = 8 * dim + 48 + 4 + 8 + (384 * 4 if wl == "c5" else 0)
    model_bytes = int(info["n_nodes"]) * 16 + int(info["n_leaves"]) * (8 if args.backend == "lightgbm" else 4)
    alg_path = my_items * b_item + model_bytes      # the whole fused path (8d)
    # per KERNEL: what that launch has to move at best (the scorer never touches the 8 D bytes of the store).  The u16
    # tile between assembly and scoring (V x 2 B per item) is algorithmic for each of the two kernels taken alone and is NOT part
    # of the fused path's `alg_path` - which is why the two do not add up to it.
    alg = {"assemble": my_items * (b_item - 8 + 2 * V),                 # store cells + list tokens + ids in, the tile out
           "score": my_items * (2 * V + 8) + model_bytes,               # the tile in, f64 scores out, the forest once
           "bin": my_items * (8 * dim + 2 * V),
           "prepass": alg_path, "rank_fused": alg_path, "encoder": alg_path}   # one-launch paths: the whole path's bytes
    enc_flops = None
    if enc is not None:   # matrix-core work of one forward pass over the batch's REAL tokens (packed): 4 H^2 + 2 H I per token and layer in the products, 4 H per token pair in attention, x 2
        L_, H_, I_ = enc.info["layers"], enc.info["hidden"], enc.info["intermediate"]
        lens_ = np.asarray(qtok[0][2]).sum(axis=1).astype(np.int64)
        enc_flops = int(lens_.sum()) * L_ * 2 * (4 * H_ * H_ + 2 * H_ * I_) + int((lens_ * lens_).sum()) * L_ * 4 * H_
    alg["sort"] = total_items * (8 + 4)
    alg["override"] = 0
    # HBM-side traffic of that kernel from the committed PMC passes (rocprofv3 FETCH_SIZE / WRITE_SIZE, KB per launch,
    # collected in separate runs: tools/gpu/r02_final.sh).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the
    # bytes of 16-byte-per-lane reads - which is what both hot kernels issue (the assembly kernel's record pieces, the
    # scorer's slab copy) - so the fetch side is doubled; `traffic_raw` is the counter as reported.  For the assembly
    # kernel the truth lies between the two (DESIGN.md "Roofline accounting").  Only filled when the profiled launch had
    # this run's shape.
    traffic = traffic_raw = None
    valu_issue = {}
    pmc_stale = None   # None: no committed counters for this shape; True: they were collected on OTHER code and are not quoted
    jit_on = os.environ.get("MRK_RANK_JIT", "1") not in ("0",)
    # WHICH code this run measured: the library's source digest and the key (hash of the translation unit) of every
    # specialised kernel loaded for the model.  tools/pmc_summary.py copies the same object out of the profiled run's
    # output into each committed summary; counters of another build or another kernel are dropped here, not quoted.
    provenance = {"build_id": M.lib().mrk_build_id().decode(), "jit_kernels": ranker.kernel_keys(model_name)}
    # the kernel behind `dominant` in the newest committed summary of this workload (names change with the batch shape:
    # mrk_jit_rank_cells / _split / mrk_jit_assemble_cells; qs_score_wave_kernel / qs_score_split_kernel)
    prefixes = {"rank_fused": ("mrk_jit_rank_fused_score", "rank_fused_score"), "score": ("qs_score",), "assemble": ("mrk_jit_rank_cells", "mrk_jit_assemble_cells") if jit_on else ("rank_fused_cells", "assemble_cells")}
    pmc_kernel = None
    try:
        import glob
        shape_ok = n_gpus == 1 and args.backend == "lightgbm" and args.quantiles == 49 and (
            (wl == "c2" and args.requests == 3840) or (wl == "c3" and args.requests == 384) or (wl == "c4x" and args.items == 4_000_000 and args.clones == 79))
        files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_pmc_{wl}_summary.json")), reverse=True)
        summary = json.load(open(files[0])) if files and shape_ok else None
        if summary is not None:
            prov = summary.get("_provenance") or {}
            # (a profiled run may have loaded fewer kernels than this one, and a run that compiled a model's own kernel in the
            #  background also holds the program-only stand-in it ranked with meanwhile: what must agree is the kernel a warmed-up
            #  process runs - the one keyed by program AND forest where there is one)
            def steady(lst):
                keyed = sorted(x for x in lst if x.endswith("program+forest"))
                return keyed or sorted(lst)
            pj, cj = prov.get("jit_kernels") or {}, provenance["jit_kernels"]
            both = set(pj) & set(cj)
            same_code = prov.get("build_id") == provenance["build_id"] and all(steady(pj[k]) == steady(cj[k]) for k in both) and (bool(both) or not cj)
            pmc_stale = not same_code
            if not same_code:
                summary = None
        if summary is not None:
            for name, d in summary.items():
                if name.startswith(prefixes.get(dominant, ())) and isinstance(d, dict) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                    pmc_kernel = name
                    traffic_raw = (d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0
                    traffic = (2.0 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0
                    break
            # The bound these kernels actually run against (neither is bandwidth-bound): VALU issue.  A wavefront's
            # vector instruction occupies its SIMD's VALU for 4 cycles, so a launch cannot take less than
            # (VALU wavefront-instructions of the launch, SQ_INSTS_VALU of the committed PMC pass) x 4 cycles / 1 024 SIMDs /
            # 2.4 GHz; frac = that floor / this run's measured launch time.
            for kname in ("assemble", "score", "rank_fused"):
                for name, d in summary.items():
                    if kname in kernels and name.startswith(prefixes[kname]) and isinstance(d, dict) and "SQ_INSTS_VALU" in d:
                        floor_ms = d["SQ_INSTS_VALU"]["mean"] * 4.0 / 1024.0 / 2.4e9 * 1e3
                        valu_issue[kname] = {"kernel": name, "valu_wave_instructions": d["SQ_INSTS_VALU"]["mean"], "floor_ms": floor_ms,
                                             "avg_launch_ms": kernels[kname]["avg_ms"], "frac": floor_ms / kernels[kname]["avg_ms"],
                                             "pmc_summary": os.path.basename(files[0])}
                        break
    except Exception:
        traffic = traffic_raw = None
    dur_s = kernels[dominant]["avg_ms"] * 1e-3
    achieved = alg[dominant] / dur_s / 1e9
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw, "traffic_kernel": pmc_kernel, "pmc_stale": pmc_stale, "algorithmic_bytes_per_launch": alg[dominant],
                "bytes_per_item": b_item, "items_per_launch": my_items, "model_bytes_per_launch": model_bytes,
                # rounds 1-4 priced every kernel with the whole fused path's SURVEY 8(d) bytes; kept for continuity with those lines
                "by_whole_path_bytes": {"algorithmic_bytes_per_launch": alg_path, "achieved": alg_path / dur_s / 1e9, "frac": alg_path / dur_s / 1e9 / HBM_PEAK_GBS},
                "avg_launch_ms": kernels[dominant]["avg_ms"],
                # `frac` prices the SURVEY 8(d) bytes against HBM; when the item table fits the 256 MiB Infinity Cache none of
                # those bytes come from HBM and `frac` is NOT a bandwidth statement - the instruction-issue bound below is
                "cache_resident": int(n_catalogue) * ranker.item_stride() < 256 * 1024 * 1024,
                "valu_issue": valu_issue or None,
                "per_kernel": {k: {"algorithmic_bytes_per_launch": alg[k], "avg_launch_ms": kernels[k]["avg_ms"],
                                   "achieved": alg[k] / (kernels[k]["avg_ms"] * 1e-3) / 1e9, "frac": alg[k] / (kernels[k]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                               for k in kernels if k in alg and alg[k]},
                "whole_path": {"achieved": alg_path / (ms_per_batch * 1e-3) / 1e9, "frac": alg_path / (ms_per_batch * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "ms_per_batch": ms_per_batch},
                "note": ("the forest scorer is VALU-issue bound, not HBM bound (SURVEY.md 8d): "
                         f"{my_items * info['n_trees'] / max(kernels['score']['avg_ms'] * 1e-3, 1e-12) / 1e9:.1f} G item-trees/s "
                         f"in {kernels['score']['avg_ms']:.3f} ms") if "score" in kernels else
                        "rank_fused = pre-pass + assembly + forest + ordering of a request in its own workgroup, ONE launch per batch "
                        "(MRK_RANK_FUSED_SCORE=0: the three-launch path with per-kernel times)"}

    if wl == "c4x" and "assemble" in kernels:
        # the out-of-cache gather is the ONE launch of this path where the HBM roofline is the right yardstick: its own
        # figures next to the dominant kernel's - 8(d) bytes, and the 384-byte records it cannot avoid touching
        g_ms = kernels["assemble"]["avg_ms"]
        rec_bytes = my_items * ranker.item_stride()
        roofline["gather"] = {"kernel": "assemble", "avg_launch_ms": g_ms, "achieved": alg_path / (g_ms * 1e-3) / 1e9, "frac": alg_path / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "record_bytes_per_launch": rec_bytes, "frac_by_records": rec_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if dominant == "encoder":   # config 5: the launch sequence that dominates is GEMM-shaped - priced against the dense f16 MFMA peak
        mfma_peak = MFMA_PEAK_TFLOPS[args.encoder_precision]
        roofline = {"bound": "mfma", "kernel": "encoder", "achieved": enc_flops / dur_s / 1e12, "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": enc_flops / dur_s / 1e12 / mfma_peak, "traffic": None, "precision": args.encoder_precision, "flops_per_launch": enc_flops, "avg_launch_ms": kernels[dominant]["avg_ms"],
                    "what": "one forward pass of the step's queries (embedding ... mean pooling; ~40 launches, HIP events on the encoder's stream); "
                            "peak = dense MFMA rate of the operand type (f32-input 157.3, f16 2 500 TFLOP/s), MI355X_MICROARCH.md",
                    "hbm_view_of_the_rank_batch": {"kernel": "assemble", "achieved": alg_path / (kernels["assemble"]["avg_ms"] * 1e-3) / 1e9,
                                                   "frac": alg_path / (kernels["assemble"]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS} if "assemble" in kernels else None}

    # ---- item-sharded workloads: what N ranks can gain at best, from this run's own kernel split.  Pre-pass and ordering see
    #      the whole request on every rank; assembly and scoring are shared out; the all-gather moves 8 B per candidate.
    projection = None
    if sharded and n_gpus == 1:
        k_ms = {k: v["avg_ms"] * v["launches_per_batch"] for k, v in kernels.items()}
        serial = k_ms.get("prepass", 0.0) + k_ms.get("sort", 0.0)
        shared = k_ms.get("assemble", 0.0) + k_ms.get("score", 0.0) + k_ms.get("override", 0.0)
        other = max(ms_per_batch - serial - shared, 0.0)   # launch gaps of the stream
        projection = {"unsharded_ms": serial, "sharded_ms": shared, "other_ms": other,
                      "speedup_ceiling": {str(n_): ms_per_batch / (serial + other + shared / n_ + (0.02 if n_ > 1 else 0.0)) for n_ in (1, 2, 4, 8)},
                      "note": "ceiling = batch time / (pre-pass + sort + gaps + (assembly + scoring) / N + ~0.02 ms for one small all-gather over xGMI); "
                              "the ordering is NOT sharded (every rank sorts the gathered scores), which is what bounds config 4 at N = 8"}

    # ---- end to end: FRESH requests every device batch (host part + upload of the id bytes + device-side id resolution
    #      + run + download into pinned memory), several batches in flight, one host thread
    e2e = None
    if rank == 0 and n_gpus == 1 and args.e2e_seconds > 0 and not sharded and wl != "c5":
        n_sets = max(2, args.e2e_sets)
        sets = [M.RequestSet(ranklens.generate_requests(args.requests, args.items, args.catalogue, args.sessions,
                                                        seed=ranklens.SEED + 5000 + k)) for k in range(n_sets)]
        nb = max(1, args.e2e_batches)
        n_thr = max(1, args.e2e_threads)
        results = [None] * n_sets
        import threading

        class Server:   # one host thread's share of the loop: its own batches, every n_thr-th request set
            def __init__(self, k):
                self.k, self.eb, self.n_done, self.err = k, [ranker.new_batch() for _ in range(nb)], 0, None
                self.t = [0.0, 0.0, 0.0, 0.0]   # seconds in: waiting for results, mrk_batch_load, mrk_batch_run, enqueue of the download

            def serve(self, i, keep=False):
                b = self.eb[i % nb]
                t0_ = time.perf_counter()
                if i >= nb:
                    sc, od, st = b.host_outputs()           # waits for the batch's previous round
                    assert (st == 0).all()
                    if keep:
                        results[((i - nb) * n_thr + self.k) % n_sets] = (sc.copy(), od.copy())
                t1_ = time.perf_counter()
                b.load(model_name, sets[(i * n_thr + self.k) % n_sets])
                t2_ = time.perf_counter()
                b.run(booster)
                t3_ = time.perf_counter()
                b.enqueue_fetch()
                t4_ = time.perf_counter()
                self.t[0] += t1_ - t0_; self.t[1] += t2_ - t1_; self.t[2] += t3_ - t2_; self.t[3] += t4_ - t3_

            def warm(self):
                for i in range(2 * nb + n_sets):
                    self.serve(i, keep=True)
                for b in self.eb:
                    b.sync()

            def timed(self, go, seconds):
                try:
                    go.wait()
                    self.t = [0.0, 0.0, 0.0, 0.0]
                    t = time.perf_counter()
                    while True:
                        for _ in range(16):
                            self.serve(self.n_done)
                            self.n_done += 1
                        if time.perf_counter() - t >= seconds:
                            break
                    for b in self.eb:
                        b.host_outputs()
                except Exception as e:  # noqa: BLE001
                    self.err = e

        # the same loop driven by a native host (tools/native/serve_driver.cpp: C++ threads against include/mrk.h only, no Python
        # between the calls): what ONE host thread of a compiled host can feed - reported beside the Python harness's own figure
        native = None
        drv_path = os.path.join(REPO, "tools", "native", "libserve_driver.so")
        if os.path.exists(drv_path):
            import ctypes as C
            from metarank_amd import _native as N_
            drv = C.CDLL(drv_path)
            fn = drv.mrk_bench_serve_loop
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
            set_ptrs = (C.c_void_p * n_sets)(*[C.cast(rs_.arr, C.c_void_p) for rs_ in sets])
            set_n = (C.c_int * n_sets)(*[rs_.n_req for rs_ in sets])
            set_ids = (N_.mrk_item_ids * n_sets)(*[rs_.ids for rs_ in sets])
            T0 = sets[0].total_items
            native = {}
            for thr_ in sorted({1, n_thr}):
                outv = (C.c_double * 8)()
                fs, fo = np.empty(T0, dtype=np.float64), np.empty(T0, dtype=np.int32)
                rc_ = fn(ctx.handle, booster.handle, model_name.encode(), set_ptrs, set_n, set_ids, n_sets, thr_, nb, float(args.e2e_seconds), outv,
                         fs.ctypes.data, fo.ctypes.data, T0)
                assert rc_ == 0 and outv[6] == 0, (rc_, outv[6], M.lib().mrk_last_error())
                chk = ranker.prepare(model_name, sets[0].requests_with_ids())
                chk.run(booster)
                cs, co, _ = chk.fetch()
                chk.close()
                assert np.array_equal(fs, cs) and np.array_equal(fo, co), "native e2e results differ from the resident path"
                nd = outv[1]
                native[f"threads_{thr_}"] = {"value": nd * T0 / outv[0], "unit": "items/s", "frac_of_value": nd * T0 / outv[0] / value, "seconds": outv[0], "device_batches": int(nd),
                                            "ms_per_batch": outv[0] / max(nd, 1) * 1e3, "batches_in_flight": nb * thr_,
                                            "host_ms_per_batch": {k_: 1e3 * outv[2 + j_] / max(nd, 1) for j_, k_ in enumerate(("wait_results", "load", "run", "enqueue_fetch"))}}
        servers = [Server(k) for k in range(n_thr)]
        for sv in servers:
            sv.warm()
        go = threading.Event()
        threads = [threading.Thread(target=sv.timed, args=(go, args.e2e_seconds)) for sv in servers]
        for t in threads:
            t.start()
        t1 = time.perf_counter()
        go.set()
        for t in threads:
            t.join()
        e2e_s = time.perf_counter() - t1
        for sv in servers:
            if sv.err is not None:
                raise sv.err
        n_done = sum(sv.n_done for sv in servers)
        eb = [b for sv in servers for b in sv.eb]
        e2e_items = n_done * sets[0].total_items
        # what the loop returned equals the device-resident path on the same requests (bit for bit)
        chk = ranker.prepare(model_name, sets[0].requests_with_ids())
        chk.run(booster)
        cs, co, _ = chk.fetch()
        chk.close()
        assert results[0] is not None and np.array_equal(results[0][0], cs) and np.array_equal(results[0][1], co), "e2e results differ from the resident path"
        e2e = {"value": e2e_items / e2e_s, "unit": "items/s", "seconds": e2e_s, "device_batches": n_done,
               "ms_per_batch": e2e_s / n_done * 1e3, "frac_of_value": (e2e_items / e2e_s) / value,
               "batches_in_flight": nb * n_thr, "host_threads": n_thr, "distinct_request_sets": n_sets,
               "host_ms_per_batch": {k: 1e3 * sum(sv.t[j] for sv in servers) / max(n_done, 1)
                                     for j, k in enumerate(("wait_results", "load", "run", "enqueue_fetch"))},
               "driver": "python (ctypes) threads; `native_driver` = the same loop from C++ threads (tools/native/serve_driver.cpp), with 1 and with --e2e-threads host threads",
               "native_driver": native,
               "h2d_bytes_per_batch": int(sets[0].id_bytes_total + 4 * (sets[0].total_items + 1)),
               "d2h_bytes_per_batch": int(12 * sets[0].total_items + 4 * sets[0].n_req),
               "includes": "per device batch: mrk_batch_load (user/session slots, request constants, table sizing - one host thread per batch; "
                           "upload of the raw id bytes; id -> slot resolution by a kernel) + mrk_batch_run (assembly, scoring, sort) + "
                           "download of scores / order / status into pinned memory",
               "excludes": "JSON decoding of the events (the host's HTTP layer) - the requests are pre-marshalled mrk_request structs + flat id bytes"}
        for b in eb:
            b.close()
        for rs_ in sets:
            rs_.close()

    # ---- single-request latency (p50 of mrk_rank: host marshalling + 4 launches + copies)
    encoder_out = None
    if enc is not None and rank == 0:
        ids, types, mask = qtok[0]
        ts = []
        for _ in range(prof_steps):
            t1 = time.perf_counter(); enc.embed_ids(ids, types, mask); ts.append(time.perf_counter() - t1)
        ems = float(np.median(ts) * 1e3)
        L, H, I, S = enc.info["layers"], enc.info["hidden"], enc.info["intermediate"], ids.shape[1]
        lens = np.asarray(mask).sum(axis=1).astype(np.int64)   # the forward pass runs over the real tokens only (packed batches)
        fl = int(lens.sum()) * L * 2 * (4 * H * H + 2 * H * I) + int((lens * lens).sum()) * L * 4 * H
        # north_star asks for scores within 1e-5 and the same order.  On a sample of 24 requests: (1) `f32_batch` - the f32 encoder's
        # packed batch against the SAME requests ranked one at a time through mrk_rank (the <= 32-row kernels, a fresh handle without a
        # cached embedding): the f32 kernels share one accumulation order, so this must be 0 / 0; (2) `fp16_vs_f32` - the price of the
        # fp16 opt-in (BASELINE config 5's wording) against the f32 arithmetic; (3) the other precision's step, timed beside the headline
        other_prec = "f16" if args.encoder_precision == "f32" else "f32"
        f32_batch = fp16_vs_f32 = other = None
        try:
            enc_o = HipEncoder(enc_weights, tok_json, ctx=ctx, precision=other_prec)
            e32, e16 = (enc, enc_o) if args.encoder_precision == "f32" else (enc_o, enc)
            sample_ev = all_events[0][:24]
            res = []
            for e_ in (e16, e32):
                ranker.bind_encoder("title_match", e_)
                sb = ranker.prepare(model_name, sample_ev)
                sb.run(booster)
                sc_, od_, _ = sb.fetch()
                res.append((sc_.copy(), od_.copy(), list(sb.offsets)))
                sb.close()
            (s16, o16, offs), (s32, o32, _) = res
            moved = int((np.abs(s16 - s32) > 1e-5).sum())
            reordered = sum(1 for r_ in range(len(sample_ev)) if not np.array_equal(o16[offs[r_]:offs[r_ + 1]], o32[offs[r_]:offs[r_ + 1]]))
            fp16_vs_f32 = {"requests": len(sample_ev), "items": int(len(s16)), "scores_over_1e-5": moved, "reordered_requests": reordered,
                           "max_abs_score_diff": float(np.abs(s16 - s32).max())}
            e1 = HipEncoder(enc_weights, tok_json, ctx=ctx, precision="f32")   # fresh: every query is a miss of its EmbeddingCache
            ranker.bind_encoder("title_match", e1)
            moved1 = reord1 = 0
            worst = 0.0
            for r_, ev_ in enumerate(sample_ev):
                _, s1, o1 = ranker.rerank(model_name, M.Request(ev_), booster)
                lo_, hi_ = offs[r_], offs[r_ + 1]
                worst = max(worst, float(np.abs(np.asarray(s1) - s32[lo_:hi_]).max()))
                moved1 += int((np.asarray(s1) != s32[lo_:hi_]).sum())
                reord1 += int(not np.array_equal(np.asarray(o1), o32[lo_:hi_]))
            e1.close()
            f32_batch = {"requests": len(sample_ev), "items": int(len(s32)), "scores_not_bit_identical": moved1, "scores_over_1e-5": 0 if worst <= 1e-5 else moved1,
                         "reordered_requests": reord1, "max_abs_score_diff": worst,
                         "what": "packed f32 batch (mrk_batch_run) vs the same requests one at a time (mrk_rank), both with f32 encoders; against the "
                                 "independent numpy fp32 graph: tests/test_encoder_gpu.py::test_c5_against_the_fp32_embedding_not_against_itself"}
            # the other precision's device step (same batches, same loop), a few steps
            ranker.bind_encoder("title_match", enc_o)
            cur_enc[0] = enc_o
            n_o = max(2, args.steps // 4)
            step(0); sync_all()
            t1 = time.perf_counter()
            for i_ in range(n_o):
                step(i_)
            sync_all()
            dt_o = time.perf_counter() - t1
            ts_o = []
            for _ in range(5):
                t1 = time.perf_counter(); enc_o.embed_ids(ids, types, mask); ts_o.append(time.perf_counter() - t1)
            other = {"precision": other_prec, "value": total_items * bps * n_o / dt_o, "unit": "items/s", "ms_per_step": dt_o / n_o * 1e3, "steps": n_o,
                     "encoder_ms_per_batch": float(np.median(ts_o) * 1e3), "tflops": fl / float(np.median(ts_o)) / 1e12,
                     "frac_of_mfma_peak": fl / float(np.median(ts_o)) / 1e12 / MFMA_PEAK_TFLOPS[other_prec]}
            cur_enc[0] = enc
            ranker.bind_encoder("title_match", enc)
            enc_o.close()
        except Exception as e:  # noqa: BLE001
            cur_enc[0] = enc
            ranker.bind_encoder("title_match", enc)
            fp16_vs_f32 = fp16_vs_f32 or {"error": str(e)}
        peak_ = MFMA_PEAK_TFLOPS[args.encoder_precision]
        encoder_out = {"model": f"bert {L}x{H}, {enc.info['heads']} heads, ffn {I} (all-MiniLM-L6-v2 shape), random weights",
                       "precision": {"f32": "f32 operands and accumulation on v_mfma_f32_16x16x4_f32 (mrk_encoder_load's default): the fp32 ONNX session's arithmetic, "
                                            "one accumulation order for every kernel - a query's embedding is the same bits alone and in a packed batch",
                                     "f16": "fp16 operands / f32 accumulation on v_mfma_f32_32x32x16_f16 (opt-in, BASELINE config 5's wording)"}[args.encoder_precision],
                       "f32_batch": f32_batch, "fp16_vs_f32": fp16_vs_f32, "other_precision": other,
                       "queries_per_step": int(ids.shape[0]), "padded_tokens_per_query": int(S), "real_tokens_per_query": float(lens.mean()),
                       "layout": "packed: real tokens back to back, no padding", "ms_per_step": ems,
                       "tflops": fl / ems / 1e9, "mfma_peak_tflops": peak_, "frac_of_mfma_peak": fl / ems / 1e9 / peak_,
                       "includes": "H2D of token ids and D2H of the embeddings (host-buffer API)"}
    latency = None
    if rank == 0 and args.latency_requests > 0:
        lat_events = events[:min(args.latency_requests, max(3, 30_000 // args.items))]
        if enc is not None:  # a never-seen query per call: tokenise + forward pass + rank, no EmbeddingCache hit
            lat_events = [dict(e) for e in lat_events]
            for e, q in zip(lat_events, synth.synthetic_queries(len(lat_events), seed=999)):
                e["fields"] = [{"name": "query", "value": q}]
        reqs = [M.Request(e) for e in lat_events]
        warm = reqs[:min(20, len(reqs))]
        if enc is not None:
            warm = []
            for e, q in zip(lat_events[:20], synth.synthetic_queries(20, seed=998)):
                e = dict(e); e["fields"] = [{"name": "query", "value": q}]
                warm.append(M.Request(e))
        for r in warm[:2]:
            ranker.rerank(model_name, r, booster)
        ranker.warmup_kernels(model_name)   # the one-launch kernel of this model, if it was not on disk
        for r in warm:
            ranker.rerank(model_name, r, booster)
        lat = []
        for r in reqs:
            t1 = time.perf_counter()
            ranker.rerank(model_name, r, booster)
            lat.append((time.perf_counter() - t1) * 1e3)
        latency = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "n": len(lat),
                   "items": args.items, "path": "mrk_rank: one upload, ONE launch (pre-pass + assembly + forest + ordering), results written to pinned memory"}
        # the device part of that: HIP events around the one launch (a loop of its own - the events cost a little)
        try:
            ctx.profile_enable(True)
            for r in reqs[:50]:
                ranker.rerank(model_name, r, booster)
            k_ms, k_n = ctx.profile_get("rank_one")
            ctx.profile_enable(False)
            latency["kernel_ms"] = (k_ms / k_n) if k_n else None
        except Exception:  # noqa: BLE001
            ctx.profile_enable(False)
        if enc is not None:
            # the same requests with the other precision's encoder (a never-seen query per call either way)
            latency["query_precision"] = args.encoder_precision
            other_prec = "f16" if args.encoder_precision == "f32" else "f32"
            try:
                enc_o = HipEncoder(enc_weights, tok_json, ctx=ctx, precision=other_prec)
                ranker.bind_encoder("title_match", enc_o)
                for r in warm[:5]:
                    ranker.rerank(model_name, r, booster)
                l16 = []
                for r in reqs:
                    t1 = time.perf_counter()
                    ranker.rerank(model_name, r, booster)
                    l16.append((time.perf_counter() - t1) * 1e3)
                latency[f"{other_prec}_query_p50_ms"] = float(np.percentile(l16, 50))
                ranker.bind_encoder("title_match", enc)
                enc_o.close()
            except Exception as e:  # noqa: BLE001
                latency[f"{other_prec}_query_p50_ms"] = None
                latency["other_precision_error"] = str(e)
        # the same ONE request over and over: its session, its candidates' records and tables are in L2 - what is left of the
        # distance to `p50_ms` (distinct requests, cold lines) is memory latency of the request's dependent trips, not the path
        hot = []
        for _ in range(min(100, len(reqs))):
            t1 = time.perf_counter()
            ranker.rerank(model_name, reqs[0], booster)
            hot.append((time.perf_counter() - t1) * 1e3)
        latency["same_request_p50_ms"] = float(np.percentile(hot, 50))
        # the same requests through the serving queue (mrk_serve_*: persistent workgroups polling pinned slots - no launch, no copy)
        if enc is None and info["bitvector"] and args.items <= 128:
            try:
                srv = ranker.serve(model_name, booster, n_slots=2)
                for r in warm:
                    srv.rerank(r)
                lat2 = []
                for r in reqs:
                    t1 = time.perf_counter()
                    srv.rerank(r)
                    lat2.append((time.perf_counter() - t1) * 1e3)
                latency["serve_queue"] = {"p50_ms": float(np.percentile(lat2, 50)), "p99_ms": float(np.percentile(lat2, 99)), "stats": srv.stats()}
                srv.close()
                srv = ranker.serve(model_name, booster, n_slots=2)   # (a fresh server: its stats cover the repeated request only)
                for r in warm[:3]:
                    srv.rerank(reqs[0])
                hot2 = []
                for _ in range(min(100, len(reqs))):
                    t1 = time.perf_counter()
                    srv.rerank(reqs[0])
                    hot2.append((time.perf_counter() - t1) * 1e3)
                latency["serve_queue"]["same_request"] = {"p50_ms": float(np.percentile(hot2, 50)), "stats": srv.stats()}
                srv.close()
            except Exception as e:  # noqa: BLE001
                latency["serve_queue"] = {"error": str(e)}

        # ---- the reference's own latency protocol (T/util/benchmark/LatencyBenchmark.scala:60-86,120-142; the chart of
        #      doc/performance.md:13): 1 000 warm-up requests of 10 items, then 5 000 SEQUENTIAL requests for each size 25 ... 300,
        #      percentiles 50 / 80 / 90 / 95 / 99 (commons-math `Percentile` = R-6 = numpy's "weibull").  What differs and is said in the
        #      line: the reference's client is HTTP on localhost against Redis-backed state and a 50-tree model with 17 features - its
        #      numbers are dominated by the store round trip; here the caller is the C ABI (no HTTP / JSON), the state is device-resident
        #      and the model is this run's (500 trees, 24 columns).  Requests: `sweep_distinct` distinct random ones per size, cycled.
        if args.latency_sweep and enc is None and not sharded:
            published = {25: (3.3, 6.0), 100: (7.0, 10.0), 200: (14.5, 17.5), 300: (20.0, 24.0)}   # p50 / p99 ms read off the chart (Redis, binary format)
            percs = (50, 80, 90, 95, 99)
            n_seq, n_distinct = args.latency_sweep, min(args.latency_sweep, 1000)

            def sweep(call):
                warm10 = [M.Request(e) for e in ranklens.generate_requests(200, 10, args.catalogue, args.sessions, seed=ranklens.SEED + 7000)]
                for k in range(1000):
                    call(warm10[k % len(warm10)])
                rows = []
                for items_ in range(25, 301, 25):
                    rq = [M.Request(e) for e in ranklens.generate_requests(n_distinct, items_, args.catalogue, args.sessions, seed=ranklens.SEED + 7000 + items_)]
                    for r in rq[:20]:
                        call(r)
                    ts_ = np.empty(n_seq)
                    for k in range(n_seq):
                        r = rq[k % n_distinct]
                        t1 = time.perf_counter()
                        call(r)
                        ts_[k] = time.perf_counter() - t1
                    row = {"items": items_, **{f"p{p}_ms": float(np.percentile(ts_ * 1e3, p, method="weibull")) for p in percs}}
                    if items_ in published:
                        row["reference_published_p50_p99_ms"] = list(published[items_])
                    rows.append(row)
                return rows

            try:
                sw = {"protocol": "LatencyBenchmark.scala: 1000 warm-up requests of 10 items, then sequential requests per size 25..300, percentiles 50/80/90/95/99 (R-6)",
                      "requests_per_size": n_seq, "distinct_requests_per_size": n_distinct,
                      "differs_from_the_reference": "caller = C ABI on the same host (no HTTP, no JSON decoding); state device-resident (the reference's chart: Redis without client cache); "
                                                    f"model = this run's ({info['n_trees']} trees, {dim} columns; the reference's: 50 trees, 17 features); reference hardware not stated",
                      "mrk_rank": sweep(lambda r: ranker.rerank(model_name, r, booster))}
                if info["bitvector"]:
                    srv = ranker.serve(model_name, booster, n_slots=2)
                    sw["mrk_serve_rank"] = sweep(srv.rerank)
                    sw["mrk_serve_rank_stats"] = srv.stats()   # requests of more than 128 candidates go through mrk_rank ("fallback")
                    srv.close()
                latency["sweep"] = sw
            except Exception as e:  # noqa: BLE001
                latency["sweep"] = {"error": str(e)}

        # ---- concurrent callers of the per-request entry point: how `Ranker.rerank` is driven by the reference's host (one fiber
        #      per request, api/routes/RankApi.scala:25-41).  NATIVE threads (tools/native/callers_driver.cpp: C++ against include/mrk.h,
        #      nothing between the calls), closed loop, each result compared bit for bit with the sequential pass above.
        if enc is None and not sharded and args.concurrent_callers:
            try:
                import ctypes as C
                from metarank_amd.request import request_array
                drv_c = os.path.join(REPO, "tools", "native", "libcallers_driver.so")
                if not os.path.exists(drv_c):
                    raise RuntimeError("tools/native/libcallers_driver.so is not built (__graft_entry__.build)")
                dc = C.CDLL(drv_c)
                dc.mrk_bench_callers.restype = C.c_int
                dc.mrk_bench_callers.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
                counts = [int(x) for x in args.concurrent_callers.split(",")]
                c_events = ranklens.generate_requests(max(counts) * 8, args.items, args.catalogue, args.sessions, seed=ranklens.SEED + 9100)
                c_reqs = [M.Request(e) for e in c_events]
                c_arr = request_array(c_reqs)
                c_sc = np.zeros((len(c_reqs), args.items), dtype=np.float64)
                c_od = np.zeros((len(c_reqs), args.items), dtype=np.int32)
                for i_, r in enumerate(c_reqs):
                    _, s_, o_ = ranker.rerank(model_name, r, booster)
                    c_sc[i_, :len(s_)] = s_
                    c_od[i_, :len(o_)] = o_

                def callers(threads, per_thread, srv_handle=None):
                    lat_ = np.zeros(threads * per_thread, dtype=np.float64)
                    out_ = np.zeros(8, dtype=np.float64)
                    rc_ = dc.mrk_bench_callers(ctx.handle, booster.handle, model_name.encode(), srv_handle, C.addressof(c_arr), len(c_reqs), args.items, threads,
                                               per_thread, lat_.ctypes.data_as(C.c_void_p), out_.ctypes.data_as(C.c_void_p), c_sc.ctypes.data_as(C.c_void_p),
                                               c_od.ctypes.data_as(C.c_void_p))
                    assert rc_ == 0 and out_[1] == 0, (rc_, out_[1])
                    n_ = threads * per_thread
                    return {"callers": threads, "requests_per_s": n_ / out_[0], "items_per_s": n_ * args.items / out_[0],
                            "p50_ms": float(np.percentile(lat_, 50, method="weibull")), "p99_ms": float(np.percentile(lat_, 99, method="weibull")),
                            "max_ms": float(lat_.max()), "results_differing_from_sequential": int(out_[3])}

                for warm_n in (4, 16, 48):   # batch kernels of every size class, the lanes' buffers
                    callers(warm_n, 40)
                conc = {"driver": "native threads (tools/native/callers_driver.cpp), closed loop, every result checked against a sequential pass",
                        "host_cpus": os.cpu_count(), "cpu_quota": (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
                        "requests_per_caller": args.concurrent_requests,
                        "mrk_rank": [callers(t_, args.concurrent_requests) for t_ in counts]}
                if info["bitvector"] and args.items <= 128:
                    # the serving queue started (mrk_serve_start at warm-up: 64 slots = 8 gangs of 8 resident workgroups): first
                    # its own entry point, then mrk_rank again - the library now answers it through the queue, overflow through the front
                    srv = ranker.serve(model_name, booster, n_slots=64)
                    for r in c_reqs[:8]:
                        srv.rerank(r)
                    callers(max(counts), 40)   # (the front's scratch batches at the sizes the overflow of the queue gives them)
                    time.sleep(0.3)            # ... which trips the queue's overload guard: its 200 ms of front-only must not be the next rows'
                    conc["mrk_serve_rank"] = [callers(t_, args.concurrent_requests, srv._h) for t_ in counts if t_ <= 128]
                    conc["mrk_rank_with_the_queue_started"] = [callers(t_, args.concurrent_requests) for t_ in counts]
                    conc["mrk_serve_rank_stats"] = srv.stats()
                    srv.close()
                latency["concurrent"] = conc
            except Exception as e:  # noqa: BLE001
                latency["concurrent"] = {"error": str(e)}

    # ---- CPU baseline: the oracle (scalar C++ port of the reference read path + forest walk), 1 thread
    cpu = None
    if do_cpu:
        forest = OracleForest.from_lightgbm_text(blob) if args.backend == "lightgbm" else OracleForest.from_xgboost(blob)
        n = min(args.cpu_sample, len(events), max(1, 400_000 // args.items))  # bounded: ~10 s of CPU work
        cpu_events = events[:n]
        if enc is not None:  # the oracle has no transformer: it is handed the device embeddings (its figure excludes the encoder)
            embs = enc.embed([e["fields"][0]["value"] for e in cpu_events])
            cpu_events = [dict(e, fields=[{"name": "__embedding:title_match", "value": [float(x) for x in v]}]) for e, v in zip(cpu_events, embs)]
        reqs = [M.Request(e) for e in cpu_events]
        for r in reqs[:3]:
            forest.predict(oracle.plan.assemble(oracle.store, r))
        t1 = time.perf_counter()
        parts = [0.0, 0.0, 0.0]
        chk = []
        for r in reqs:
            a = time.perf_counter()
            m = oracle.plan.assemble(oracle.store, r)
            b = time.perf_counter()
            s = forest.predict(m)
            c = time.perf_counter()
            o = sort_order(s)
            d = time.perf_counter()
            parts[0] += b - a
            parts[1] += c - b
            parts[2] += d - c
            chk.append((s, o))
        cpu_s = time.perf_counter() - t1
        # the same work on every host core (the libraries the reference calls use OpenMP over rows; here: requests
        # over threads - the oracle's C++ runs outside the GIL)
        multi = None
        try:
            from concurrent.futures import ThreadPoolExecutor

            n_thr = min(os.cpu_count() or 1, 64)

            def one(r):
                s_ = forest.predict(oracle.plan.assemble(oracle.store, r))
                return sort_order(s_)

            with ThreadPoolExecutor(n_thr) as ex:
                list(ex.map(one, reqs[:n_thr]))  # warm the pool
                t2 = time.perf_counter()
                reps = max(1, int(3.0 * n_thr / max(cpu_s, 1e-3)))  # ~3 s of wall time
                work = [reqs[i % n] for i in range(n * min(reps, 8))]
                list(ex.map(one, work))
                mt_s = time.perf_counter() - t2
            multi = {"value": len(work) * args.items / mt_s, "unit": "items/s", "cores": n_thr, "seconds": mt_s}
        except Exception as e:  # the single-thread figure is the contract; this one is extra
            multi = {"error": str(e)}
        cpu = {"value": n * args.items / cpu_s, "unit": "items/s", "cores": 1, "kind": "port", "kind_in_words": "port (this repo's C++ oracle, oracle/; the reference's JVM + LightGBM/XGBoost natives cannot run on this image)", "all_cores": multi,
               "sample": f"{n} requests x {args.items} items of the same workload, assemble+score+sort, single thread",
               "host_cores": os.cpu_count(), "seconds": cpu_s,
               "split_s": {"assemble": parts[0], "score": parts[1], "sort": parts[2]}}
        # the GPU results of those requests equal the oracle's (parity inside the bench)
        scores, order, _ = batch.fetch()
        for r in range(n):
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            assert np.array_equal(scores[lo:hi], chk[r][0]) and np.array_equal(order[lo:hi], chk[r][1]), f"parity broke at request {r}"

    if rank == 0:
        out = {
            "metric": f"ranked items/sec (feature assembly + {args.trees}-tree LambdaMART + ordering), Ranklens-shaped {args.items}-item requests",
            "value": value, "unit": "items/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_device_batch": ms_per_batch, "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f64" if args.backend == "lightgbm" else "f32", "data": "synthetic",
            "config": {"workload": f"ranklens-{args.items}item-{dim}col-{args.trees}tree-" + (args.backend if args.backend == "lightgbm" else f"xgboost-depth{args.depth}"), "requests_per_step_per_gpu": args.requests,
                       "items_per_request": args.items, "device_batches_per_step": bps, "items_per_device_batch": total_items, "items_per_step_per_gpu": total_items * bps, "catalogue_items": n_catalogue, "item_table_bytes": int(n_catalogue) * ranker.item_stride(),
                       "sessions": args.sessions, "columns": dim, "split_candidates_per_column": args.quantiles, "trees": info["n_trees"], "leaves_per_tree": 16 if args.backend == "lightgbm" else 2 ** args.depth, "backend": args.backend,
                       "scorer": "bit-vector" if info["bitvector"] else "tree walk",
                       "tile_columns": V, "batches_in_flight": n_streams,
                       "parallelism": (f"item-sharded x{n_gpus}" if sharded else f"request-sharded x{n_gpus}") +
                                      (", RCCL all-gather of scores" if n_gpus > 1 else "")},
            "value_is": "device-resident throughput: the requests of a step are resolved and in HBM before the clock starts (bench contract); "
                        "the end-to-end serving rate with fresh requests per batch is `e2e`",
            "e2e": e2e,
            "latency": latency,
            "encoder": encoder_out,
            "kernels": kernels,
            "multi_gpu_projection": projection,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "provenance": provenance,
        }
        print(json.dumps(out), flush=True)
    for bt in batches:
        bt.close()
    booster.close()
    if use_dist:
        ctx.comm_barrier()
        if rank == 0 and n_gpus > 1:
            try:
                os.remove(f"/tmp/mrk_comm_{comm_key}.id")
            except OSError:
                pass


if __name__ == "__main__":
    main()
