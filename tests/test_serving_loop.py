"""The serving loop of the C ABI (mrk_batch_create / mrk_batch_load with flat item ids / mrk_batch_run /
mrk_batch_enqueue_fetch / mrk_batch_host_outputs): item ids travel as UTF-8 bytes and are resolved to store slots by a
kernel (csrc/resolve.hip), pre-pass tables are sized from bounds, results land in pinned memory.  Everything must equal
the oracle - and the pointer-style path (host lookups, exact table sizes) - bit for bit."""
import threading
import time

import numpy as np
import pytest

import metarank_amd as M
from backends import HipBackend, OracleBackend
from workloads import ranklens, synth

N_ITEMS, N_SESS = 3000, 300


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.fixture(scope="module")
def oracle_c2():
    b = OracleBackend(ranklens.ranklens_config(), "xgboost")
    ranklens.load_state(b, ranklens.generate_state(N_ITEMS, N_SESS))
    return b


def _requests():
    reqs = ranklens.generate_requests(40, 100, N_ITEMS, N_SESS, seed=71)
    reqs += ranklens.generate_requests(3, 1, N_ITEMS, N_SESS, seed=72) + ranklens.generate_requests(2, 300, N_ITEMS, N_SESS, seed=73)
    # ids the store has never seen, an id that is a prefix / an extension of a known one, an empty id, non-ASCII bytes
    reqs.append({"id": "odd", "timestamp": ranklens.TS, "user": None, "session": reqs[0]["session"], "fields": [],
                 "items": [{"id": "nobody"}, {"id": "12"}, {"id": "1"}, {"id": "123456789"}, {"id": ""}, {"id": "фильм-7"}, {"id": "12"}]})
    reqs.append({"id": "none", "timestamp": ranklens.TS, "user": None, "session": None, "fields": [], "items": []})
    reqs += ranklens.generate_requests(2, 64, N_ITEMS, N_SESS, seed=74)
    return reqs


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [True, False])
def test_flat_ids_are_resolved_on_the_device_and_match_the_oracle(oracle_c2, pinned):
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
        reqs = _requests()
        mats = [oracle_c2.matrix(ev) for ev in reqs]
        q = ranklens.column_quantiles(np.concatenate([m for m in mats if len(m)]))
        blob = synth.synthetic_lgbm_model(n_trees=300, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.05, missing="per_feature")
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        expected = [oracle_c2.rerank(ev) for ev in reqs]
        rs = M.RequestSet(reqs, pinned=pinned)
        batch = hip.ranker.new_batch()
        for rep in range(2):  # the second pass reuses every buffer of the batch
            batch.load("xgboost", rs)
            batch.run(hip.booster)
            batch.enqueue_fetch()
            scores, order, status = batch.host_outputs()
            assert (status == 0).all()
            for r, (_, es, eo) in enumerate(expected):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                assert same(scores[lo:hi], es), (rep, r)
                assert order[lo:hi].tolist() == eo.tolist(), (rep, r)
        # the explain matrix of the same batch is the oracle's (slots resolved by the device feed the matrix path too)
        _, _, mat = batch.fetch(matrix=True)
        for r in range(len(reqs)):
            assert same(mat[batch.offsets[r]:batch.offsets[r + 1]], mats[r]), r
        # and the pointer-style load of the same requests into the same batch object gives the same bytes
        s1 = np.array(scores)
        o1 = np.array(order)
        batch.load("xgboost", reqs)
        batch.run(hip.booster)
        s2, o2, _ = batch.fetch()
        assert same(s1, s2) and (o1 == o2).all()
        # a smaller and a larger request set through the same batch
        for sub in (reqs[:3], reqs + reqs):
            rs2 = M.RequestSet(sub, pinned=pinned)
            batch.load("xgboost", rs2)
            batch.run(hip.booster)
            scores, order, status = batch.host_outputs()  # no enqueue_fetch: host_outputs asks for the download itself
            exp = [oracle_c2.rerank(ev) for ev in sub]
            for r, (_, es, eo) in enumerate(exp):
                lo, hi = batch.offsets[r], batch.offsets[r + 1]
                assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), r
            rs2.close()
        batch.close()
        rs.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_items_put_after_the_first_load_are_found(oracle_c2):
    """The device mirror of the id table follows the store: a few new ids (entry-wise update), then enough to make the
    host table rehash (whole-table update)."""
    cfg = ranklens.ranklens_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        for be in (orc, hip):
            ranklens.load_state(be, ranklens.generate_state(500, 50))
        batch = hip.ranker.new_batch()
        n_known = 500
        for extra in (3, 2000):
            for be in (orc, hip):
                for i in range(n_known, n_known + extra):
                    be.put_double(f"item={i}/popularity", float(i % 97))
                    be.put_string_list(f"item={i}/divers_genres", ["drama", "noir"][: 1 + i % 2])
            n_known += extra
            reqs = ranklens.generate_requests(6, 80, n_known, 50, seed=90 + extra)
            rs = M.RequestSet(reqs)
            batch.load("xgboost", rs)
            batch.run(None)  # NoopModel: the matrix is what is under test
            _, _, mat = batch.fetch(matrix=True)
            for r, ev in enumerate(reqs):
                assert same(mat[batch.offsets[r]:batch.offsets[r + 1]], orc.matrix(ev)), (extra, r)
            rs.close()
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_batches_in_flight_from_several_threads_with_puts_in_between(oracle_c2):
    """Three host threads drive their own batches (load -> run -> fetch) while a fourth keeps putting values under keys
    the requests do not read; every result equals the oracle's."""
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
        sets = [ranklens.generate_requests(64, 100, N_ITEMS, N_SESS, seed=80 + k) for k in range(3)]
        q = ranklens.column_quantiles(np.concatenate([oracle_c2.matrix(ev) for ev in sets[0][:16]]))
        blob = synth.synthetic_lgbm_model(n_trees=100, n_features=24, quantiles=q)
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        expected = [[oracle_c2.rerank(ev) for ev in reqs] for reqs in sets]
        errors = []
        stop = threading.Event()

        def writer():
            i = 0
            while not stop.is_set():
                hip.put_double(f"item=new-{i % 50}/popularity", float(i))  # new items: the id table grows under the readers
                i += 1

        def serve(k):
            try:
                rs = M.RequestSet(sets[k])
                batch = hip.ranker.new_batch()
                for _ in range(12):
                    batch.load("xgboost", rs)
                    batch.run(hip.booster)
                    batch.enqueue_fetch()
                    scores, order, status = batch.host_outputs()
                    assert (status == 0).all()
                    for r, (_, es, eo) in enumerate(expected[k]):
                        lo, hi = batch.offsets[r], batch.offsets[r + 1]
                        assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist(), (k, r)
                batch.close()
                rs.close()
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        w = threading.Thread(target=writer)
        w.start()
        threads = [threading.Thread(target=serve, args=(k,)) for k in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        stop.set()
        w.join()
        assert not errors, errors
    finally:
        hip.close()


def test_id_hash_of_the_device_equals_the_hosts(tmp_path):
    """CPU: id_hash_bytes (device_types.hpp, what resolve.hip computes) == SlotMap::hash (store.hpp, what the host's table
    was built with) on ids of every length 0..40 and random bytes."""
    import os
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.cpp"
    src.write_text('''
#include <cstdio>
#include <random>
#include <string>
#include "store.hpp"
int main() {
  std::mt19937_64 g(7);
  for (int len = 0; len <= 40; ++len)
    for (int rep = 0; rep < 200; ++rep) {
      std::string s(len, 0);
      for (auto &c : s) c = (char)(g() & 0xff);
      if (mrk::SlotMap::hash(s.data(), s.size()) != mrk::id_hash_bytes((const uint8_t *)s.data(), s.size())) { printf("MISMATCH len %d\\n", len); return 1; }
    }
  printf("ok\\n");
  return 0;
}
''')
    exe = tmp_path / "h"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", str(src), "-o", str(exe),
                           "-I" + os.path.join(repo, "metarank_amd", "csrc")])
    assert subprocess.check_output([str(exe)], text=True).strip() == "ok"


@pytest.mark.gpu
def test_cloned_items_rank_like_their_originals():
    """mrk_debug_clone_items (how bench.py --workload c4x grows the catalogue past the Infinity Cache): a clone has its
    original's state, so inside one request its matrix row equals the original's - a property that holds at any size."""
    cfg = ranklens.ranklens_config()
    orc, hip = OracleBackend(cfg, "xgboost"), HipBackend(cfg, "xgboost")
    try:
        for be in (orc, hip):
            ranklens.load_state(be, ranklens.generate_state(400, 30))
        n = hip.ranker.clone_items(3)
        assert n >= 1600
        info = hip.ranker.store_info(1)
        assert info["stride"] % 128 == 0 and info["slots"] == n
        ev = ranklens.generate_requests(1, 60, 400, 30, seed=5)[0]
        originals = [it["id"] for it in ev["items"]]
        ev["items"] = [{"id": i} for i in originals] + [{"id": f"{i}#{1 + k % 3}"} for k, i in enumerate(originals)]
        batch = hip.ranker.new_batch()
        batch.load("xgboost", M.RequestSet([ev]))
        batch.run(None)
        _, _, mat = batch.fetch(matrix=True)
        assert same(mat[:60], mat[60:])
        # and the originals' rows are the oracle's for the same 120-candidate request with the clones unknown to it:
        # per-item columns only (diversity / interacted_with see 120 candidates on the device, the oracle sees the
        # clones as items without state)
        om = orc.matrix(ev)
        per_item = [c for c in range(24) if c not in (15, 16, 17, 18, 19)]
        assert same(mat[:60][:, per_item], om[:60][:, per_item])
        batch.close()
    finally:
        hip.close()


@pytest.mark.gpu
def test_rccl_code_path_with_a_world_of_one(oracle_c2):
    """The library's own RCCL communicator (csrc/comm.cpp) on one GPU: ncclGetUniqueId -> ncclCommInitRank(world 1) ->
    mrk_batch_run_sharded (slice, in-place ncclAllGather of the scores, sort) and mrk_batch_gather_scores give exactly
    what the plain run gives; the host-value collectives are the identity."""
    ctx = M.Context(0)
    hip = HipBackend(ranklens.ranklens_config(), "xgboost", ctx)
    try:
        ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
        assert ctx.comm_world == 1 and ctx.comm_rank == 0
        uid = M.Context.comm_unique_id()
        assert len(uid) == 128
        ctx.comm_init(uid, 0, 1)
        assert ctx.comm_world == 1 and ctx.comm_rank == 0
        assert ctx.comm_max(2.5) == 2.5
        ctx.comm_barrier()
        with pytest.raises(M.MrkError):
            ctx.comm_init(uid, 0, 1)   # one communicator per context
        reqs = ranklens.generate_requests(5, 100, N_ITEMS, N_SESS, seed=91) + ranklens.generate_requests(1, 6000, N_ITEMS, N_SESS, seed=92)
        q = ranklens.column_quantiles(np.concatenate([oracle_c2.matrix(ev) for ev in reqs[:5]]))
        blob = synth.synthetic_lgbm_model(n_trees=120, n_features=24, quantiles=q)
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        batch = hip.ranker.prepare("xgboost", reqs)
        batch.run(hip.booster)
        s0, o0, _ = batch.fetch()
        b2 = hip.ranker.prepare("xgboost", reqs)
        b2.run_sharded(hip.booster)
        s1, o1, _ = b2.fetch()
        assert same(s0, s1) and (o0 == o1).all()
        for r, ev in enumerate(reqs):
            _, es, eo = oracle_c2.rerank(ev)
            lo, hi = b2.offsets[r], b2.offsets[r + 1]
            assert same(s1[lo:hi], es) and o1[lo:hi].tolist() == eo.tolist(), r
        # the explicit three-step form, and the replica merge
        b2.run_shard(hip.booster, ctx.comm_rank, ctx.comm_world)
        b2.allgather_scores()
        b2.sort()
        s2, o2, _ = b2.fetch()
        assert same(s0, s2) and (o0 == o2).all()
        assert b2.gather_scores() not in (None, 0)   # ncclAllGather into the batch's merge buffer (one rank: a copy)
        b2.sync()
        assert (b2.status() == 0).all()
        # per-request failures survive the merge: the status words of every rank are all-gathered and OR-ed before the
        # sort (an item that fails its request lies in ONE rank's slice; here: the normalised rate's / by zero)
        hip.put_periodic("global/ctr_click_norm", [0, 5])
        b3 = hip.ranker.prepare("xgboost", reqs)
        b3.run_sharded(hip.booster)
        assert (b3.status() == -5).all()   # MRK_ERR_ARITHMETIC
        b3.close()
        batch.close()
        b2.close()
    finally:
        hip.close()
        ctx.close()


def _drive(hip, reqs, rounds, out, errs, k):
    """one host thread's serving loop on ITS context: single requests through mrk_rank and device batches, interleaved"""
    try:
        for r in range(rounds):
            batch = hip.ranker.prepare("xgboost", reqs)
            batch.run(hip.booster)
            for ev in reqs[:6]:
                _, s1, o1 = hip.ranker.rerank("xgboost", ev, hip.booster)
                out.setdefault((k, "one", ev["id"]), []).append((s1.copy(), o1.copy()))
            s, o, _ = batch.fetch()
            assert (batch.status() == 0).all()
            out.setdefault((k, "batch"), []).append((s.copy(), o.copy(), list(batch.offsets)))
            batch.close()
    except Exception as e:  # noqa: BLE001
        errs.append((k, e))


@pytest.mark.gpu
def test_two_contexts_in_one_process_on_separate_threads(oracle_c2):
    """HipConfig(devices: List[Int]) inside ONE host process (SURVEY 8b touch point 1; M/config/BoosterConfig.scala:96-104, the
    reference's host is one JVM): mrk_init with two ordinals returns two contexts - here both on device 0, which is what one
    GPU box can show of the threading -, each with its own store replica, model and specialised kernels.  Two host threads
    drive mrk_rank and device batches on their context CONCURRENTLY while a third thread keeps writing to both stores
    (state of items no request names, so results stay comparable): every result of either context equals the oracle's,
    bit for bit, and nothing deadlocks.  mrk_comm_init_local: a world of one works, two ranks on one GPU are refused."""
    assert M.Context.device_count() >= 1
    with pytest.raises(M.MrkError):
        M.Context.create_many([0, M.Context.device_count()])      # out of range: nothing is created
    ctxs = M.Context.create_many([0, 0])
    assert len(ctxs) == 2 and ctxs[0].handle.value != ctxs[1].handle.value
    hips = [HipBackend(ranklens.ranklens_config(), "xgboost", c) for c in ctxs]
    try:
        for h in hips:
            ranklens.load_state(h, ranklens.generate_state(N_ITEMS, N_SESS))
        reqs = ranklens.generate_requests(24, 100, N_ITEMS, N_SESS, seed=171) + ranklens.generate_requests(2, 700, N_ITEMS, N_SESS, seed=172)
        q = ranklens.column_quantiles(np.concatenate([oracle_c2.matrix(ev) for ev in reqs[:6]]))
        blob = synth.synthetic_lgbm_model(n_trees=150, n_features=24, quantiles=q)
        oracle_c2.load_model(blob, 0)
        for h in hips:
            h.load_model(blob, 0)
            h.ranker.rerank("xgboost", reqs[0], h.booster)     # first use: kernels compiled / loaded per context
        want = [oracle_c2.rerank(ev) for ev in reqs]
        out, errs, stop = {}, [], threading.Event()

        def writer():
            i = 0
            try:
                while not stop.is_set():
                    for h in hips:
                        h.put_double(f"item=fresh{i % 4000}/popularity", float(i))       # new slots, dirty ranges, table growth -> flushes under the readers
                        h.put_string_list(f"item=fresh{i % 4000}/genre", ["drama", "comedy"][: 1 + i % 2])
                    i += 1
            except Exception as e:  # noqa: BLE001
                errs.append(("writer", e))

        threads = [threading.Thread(target=_drive, args=(hips[k], reqs, 6, out, errs, k)) for k in range(2)] + [threading.Thread(target=writer)]
        for t in threads:
            t.start()
        for t in threads[:2]:
            t.join(timeout=300)
        stop.set()
        threads[2].join(timeout=60)
        assert not any(t.is_alive() for t in threads), "a serving thread did not finish"
        assert not errs, errs
        for k in range(2):
            for s, o, offs in out[(k, "batch")]:
                for r in range(len(reqs)):
                    lo, hi = offs[r], offs[r + 1]
                    assert same(s[lo:hi], want[r][1]) and o[lo:hi].tolist() == want[r][2].tolist(), (k, r)
            for r, ev in enumerate(reqs[:6]):
                for s1, o1 in out[(k, "one", ev["id"])]:
                    assert same(s1, want[r][1]) and o1.tolist() == want[r][2].tolist(), (k, r)
        # communicators of contexts living in one process
        with pytest.raises(M.MrkError):
            M.Context.comm_init_local(ctxs)                      # two ranks on ONE GPU: RCCL wants one rank per device
        assert ctxs[0].comm_world == 1 and ctxs[1].comm_world == 1
        M.Context.comm_init_local(ctxs[:1])                      # a world of one
        assert ctxs[0].comm_world == 1 and ctxs[0].comm_rank == 0 and ctxs[0].comm_max(1.5) == 1.5
        b = hips[0].ranker.prepare("xgboost", reqs)
        b.run_sharded(hips[0].booster)
        s, o, _ = b.fetch()
        for r in range(len(reqs)):
            lo, hi = b.offsets[r], b.offsets[r + 1]
            assert same(s[lo:hi], want[r][1]) and o[lo:hi].tolist() == want[r][2].tolist(), r
        b.close()
    finally:
        for h in hips:
            h.close()
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_item_sharded_rank_over_two_devices_of_one_process(oracle_c2):
    """mrk_comm_init_local over two GPUs of this process (skipped on a one-GPU box): each context's thread runs
    mrk_batch_run_sharded on its replica - slice, in-place ncclAllGather, sort - and BOTH ranks end with the unsharded
    oracle's scores and order for a 6 000-candidate request."""
    if M.Context.device_count() < 2:
        pytest.skip("needs two GPUs in one process")
    ctxs = M.Context.create_many([0, 1])
    hips = [HipBackend(ranklens.ranklens_config(), "xgboost", c) for c in ctxs]
    try:
        for h in hips:
            ranklens.load_state(h, ranklens.generate_state(N_ITEMS, N_SESS))
        M.Context.comm_init_local(ctxs)
        assert [c.comm_rank for c in ctxs] == [0, 1] and all(c.comm_world == 2 for c in ctxs)
        reqs = ranklens.generate_requests(1, 6000, N_ITEMS, N_SESS, seed=192) + ranklens.generate_requests(3, 100, N_ITEMS, N_SESS, seed=193)
        q = ranklens.column_quantiles(np.concatenate([oracle_c2.matrix(ev) for ev in reqs[1:]]))
        blob = synth.synthetic_lgbm_model(n_trees=120, n_features=24, quantiles=q)
        oracle_c2.load_model(blob, 0)
        for h in hips:
            h.load_model(blob, 0)
        got, errs = [None, None], []

        def rank_thread(k):
            try:
                b = hips[k].ranker.prepare("xgboost", reqs)
                b.run_sharded(hips[k].booster)
                s, o, _ = b.fetch()
                got[k] = (s.copy(), o.copy(), list(b.offsets))
                b.close()
            except Exception as e:  # noqa: BLE001
                errs.append((k, e))

        ts = [threading.Thread(target=rank_thread, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not errs and not any(t.is_alive() for t in ts), errs
        for k in range(2):
            s, o, offs = got[k]
            for r, ev in enumerate(reqs):
                _, es, eo = oracle_c2.rerank(ev)
                assert same(s[offs[r]:offs[r + 1]], es) and o[offs[r]:offs[r + 1]].tolist() == eo.tolist(), (k, r)
    finally:
        for h in hips:
            h.close()
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_item_id_offsets_from_the_wire_are_checked(oracle_c2):
    """mrk_item_ids.offsets is untrusted input: a last offset past bytes_len refuses the whole load; an item whose
    offsets descend (or pass bytes_len) fails ITS request with MRK_ERR_INVALID_ARG, the other requests rank as usual -
    and nothing reads outside the uploaded bytes."""
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    try:
        ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
        reqs = ranklens.generate_requests(4, 50, N_ITEMS, N_SESS, seed=75)
        blob = synth.synthetic_lgbm_model(n_trees=40, n_features=24, seed=2)
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        rs = M.RequestSet(reqs, pinned=False)
        batch = hip.ranker.new_batch()
        good = rs._offs.copy()
        # (1) the last offset passes bytes_len
        rs._offs[-1] = good[-1] + 1000
        with pytest.raises(M.MrkError) as ei:
            batch.load("xgboost", rs)
        assert ei.value.status == -1
        # (2) descending offsets inside request 2, and a huge one inside request 3 (the last offset itself is fine)
        rs._offs[:] = good
        rs._offs[2 * 50 + 7] = good[2 * 50 + 7 + 1] + 5      # > its successor: item 2*50+7 has o1 < o0
        rs._offs[3 * 50 + 10] = 0xfffffff0                    # far past bytes_len
        batch.load("xgboost", rs)
        batch.run(hip.booster)
        scores, order, status = batch.host_outputs()
        assert status.tolist() == [0, 0, -1, -1]
        for r in (0, 1):
            _, es, eo = oracle_c2.rerank(reqs[r])
            lo, hi = batch.offsets[r], batch.offsets[r + 1]
            assert same(scores[lo:hi], es) and order[lo:hi].tolist() == eo.tolist()
        # (3) and the batch recovers with good offsets
        rs._offs[:] = good
        batch.load("xgboost", rs)
        batch.run(hip.booster)
        _, _, status = batch.host_outputs()
        assert (status == 0).all()
        batch.close()
        rs.close()
    finally:
        hip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["mrk_rank", "mrk_serve_rank", "mrk_rank_with_a_queue"])
def test_native_callers_stress_the_front_while_a_writer_puts(oracle_c2, tmp_path, entry):
    """(`mrk_rank_with_a_queue`: the callers use mrk_rank while a serving queue of the model is started - the library routes
    them through the queue's 64 slots, 8 gangs of 8 resident workgroups, and the overflow through the front.)
    64 NATIVE threads (tools/native/callers_driver.cpp: C++ against include/mrk.h, no interpreter between the calls) x 150
    requests each through mrk_rank's pipelined front - several lanes in flight, ids of combined batches resolved on the device -
    and through the serving queue, while a writer keeps putting values (the id table grows; every put needs the store
    exclusively, and must not starve behind the overlapping leaders).  Every result equals the oracle's bit for bit; the
    writer makes progress; no call takes seconds."""
    import ctypes as C
    import os
    import subprocess

    from metarank_amd import _native as N
    from metarank_amd.request import request_array

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libcallers_driver.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-I", os.path.join(repo, "include"),
                           os.path.join(repo, "tools", "native", "callers_driver.cpp"), "-o", so, "-L", os.path.dirname(N.LIB_PATH), "-lmrk_hip",
                           "-Wl,-rpath," + os.path.dirname(N.LIB_PATH), "-pthread"])
    hip = HipBackend(ranklens.ranklens_config(), "xgboost")
    srv = None
    try:
        ranklens.load_state(hip, ranklens.generate_state(N_ITEMS, N_SESS))
        events = ranklens.generate_requests(120, 100, N_ITEMS, N_SESS, seed=91) + ranklens.generate_requests(8, 7, N_ITEMS, N_SESS, seed=92)
        events.append({"id": "odd", "timestamp": ranklens.TS, "user": None, "session": events[0]["session"], "fields": [],
                       "items": [{"id": "nobody"}, {"id": "12"}, {"id": "1"}, {"id": "123456789"}, {"id": ""}, {"id": "фильм-7"}, {"id": "12"}]})
        q = ranklens.column_quantiles(np.concatenate([oracle_c2.matrix(ev) for ev in events[:16]]))
        blob = synth.synthetic_lgbm_model(n_trees=150, n_features=24, quantiles=q, cat_features=[7], cat_prob=0.02)
        oracle_c2.load_model(blob, 0)
        hip.load_model(blob, 0)
        items = 100
        sc = np.zeros((len(events), items), dtype=np.float64)
        od = np.zeros((len(events), items), dtype=np.int32)
        for i, ev in enumerate(events):
            _, es, eo = oracle_c2.rerank(ev)
            sc[i, :len(es)] = es
            od[i, :len(eo)] = eo
        reqs = [M.Request(e) for e in events]
        arr = request_array(reqs)
        for r in reqs[:4]:
            hip.ranker.rerank("xgboost", r, hip.booster)
        hip.ranker.warmup_kernels("xgboost")
        if entry != "mrk_rank":
            srv = hip.ranker.serve("xgboost", hip.booster, n_slots=64)
            for r in reqs[:4]:
                srv.rerank(r)
        N.lib()
        d = C.CDLL(so)
        d.mrk_bench_callers.restype = C.c_int
        d.mrk_bench_callers.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        def drive(threads, per_thread):
            lat = np.zeros(threads * per_thread, dtype=np.float64)
            out = np.zeros(8, dtype=np.float64)
            rc = d.mrk_bench_callers(hip.ranker.ctx.handle, hip.booster.handle, b"xgboost", srv._h if entry == "mrk_serve_rank" else None, C.addressof(arr), len(reqs),
                                     items, threads, per_thread, lat.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                     od.ctypes.data_as(C.c_void_p))
            return rc, lat, out

        # warm-up: combined batches of every size class reach kernels single requests never use (under MRK_RANK_JIT=1 - what the
        # suite runs with - the first use of each waits ~4 s for the compiler); their results are checked too
        for warm_threads in (3, 12, 40):
            rc, _, out = drive(warm_threads, 6)
            assert rc == 0 and out[1] == 0 and out[3] == 0, (warm_threads, rc, out[1], out[3])
        stop = threading.Event()
        puts = [0]

        def writer():   # ~5 000 puts a second: above what the reference's import path sustains (doc/performance.md:7: 1 000 - 3 000 events/s)
            i = 0
            while not stop.is_set():
                hip.put_double(f"item=stress-{i % 500}/popularity", float(i))  # items no request names
                i += 1
                puts[0] = i
                time.sleep(0.0002)

        w = threading.Thread(target=writer)
        w.start()
        threads, per_thread = 64, 150
        rc, lat, out = drive(threads, per_thread)
        stop.set()
        w.join()
        print(f"\n{entry}: {threads * per_thread / out[0]:.0f} requests/s with a writer ({puts[0]} puts), p50 {np.percentile(lat, 50):.3f} ms, "
              f"p99 {np.percentile(lat, 99):.3f} ms, max {lat.max():.1f} ms")
        assert rc == 0 and out[1] == 0, (rc, out[1])
        assert out[3] == 0, f"{int(out[3])} of {threads * per_thread} concurrent results differ from the oracle"
        assert puts[0] > 20, "the writer starved behind the readers"
        assert lat.max() < 2000.0
        if srv is not None:
            st = srv.stats()
            print(f"   queue: {st}")
            assert st["queue"] > threads * per_thread // 4, st   # the queue, not its fallback, did the work
    finally:
        if srv is not None:
            srv.close()
        hip.close()
