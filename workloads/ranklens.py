"""Synthetic Ranklens-shaped workload: the stock Ranklens model's 24 columns
(config C2) and the 64-column mixed config (C3), a deterministic catalogue / counter / session
state generator and request generator.

The real Ranklens events and models are Git-LFS pointers in the reference and there is no network,
so everything here is generated: numpy Generator(PCG64(20250718)).  State is emitted as a stream
of (kind, key, value) puts in Metarank's own key encoding (Key.encode, model/Key.scala:9) so the
same stream loads the device store (through the C ABI) and the CPU oracle.
"""
from __future__ import annotations

import numpy as np

SEED = 20250718
GENRES = ["drama", "comedy", "thriller", "action", "adventure", "romance", "crime", "science fiction", "fantasy",
          "family", "horror", "mystery", "animation", "history", "music", "war", "western", "documentary", "tv movie",
          "foreign"]
TS = 1661345221008  # run_quickstart.sh:41


def ranklens_config() -> dict:
    """The stock Ranklens ranking model (24 columns): same features, order and parameters as the
    reference's src/test/resources/ranklens/config.yml model `xgboost` (LightGBM backend, 500 iterations)."""
    num = lambda n: {"name": n, "type": "number", "scope": "item", "source": f"metadata.{n}"}
    rate = lambda n, **kw: dict({"name": n, "type": "rate", "top": "click", "bottom": "impression", "bucket": "24h",
                                 "periods": [7, 30]}, **kw)
    div = lambda n, src: {"name": n, "type": "diversity", "source": f"item.{src}"}
    features = [
        rate("ctr_tag", scope="item.tag"), rate("ctr_genre", scope="item.genre"),
        {"name": "position", "type": "position", "position": 5},
        num("popularity"), num("vote_avg"), num("vote_cnt"), num("budget"), num("release_date"), num("runtime"),
        {"name": "title_length", "type": "word_count", "source": "metadata.title", "scope": "item"},
        {"name": "genre", "type": "string", "scope": "item", "source": "metadata.genres", "encode": "index", "values": GENRES[:15]},
        rate("ctr", normalize={"weight": 10}),
        {"name": "profile", "type": "interacted_with", "interaction": "click",
         "field": ["item.genres", "item.actors", "item.tags", "item.director"], "scope": "session", "count": 100, "duration": "24h"},
        div("divers_genres", "genres"), div("divers_actors", "actors"), div("divers_tags", "tags"),
        div("divers_year", "release_date"), div("divers_popularity", "popularity"),
        {"name": "visitor_click_count", "type": "interaction_count", "interaction": "click", "scope": "session"},
        {"name": "global_item_click_count", "type": "interaction_count", "interaction": "click", "scope": "item"},
        {"name": "day_item_click_count", "type": "window_count", "interaction": "click", "scope": "item", "bucket": "24h", "periods": [7, 30]},
    ]
    model_features = ["popularity", "vote_avg", "vote_cnt", "budget", "release_date", "runtime", "title_length", "genre", "ctr",
                      "profile", "position", "divers_genres", "divers_actors", "divers_tags", "divers_year", "divers_popularity",
                      "ctr_tag", "ctr_genre"]
    return {"features": features,
            "models": {"xgboost": {"type": "lambdamart", "backend": {"type": "lightgbm", "iterations": 500}, "features": model_features}}}


def c3_config() -> dict:
    """C3: 64 mixed columns = the 24 Ranklens columns + 2 window_count (P=2) + interaction_count +
    20 extra numbers + vector(dim 8) + string onehot(dim 7)."""
    cfg = ranklens_config()
    extra = [{"name": f"x{i}", "type": "number", "scope": "item", "source": f"metadata.x{i}"} for i in range(20)]
    extra.append({"name": "week_item_impressions", "type": "window_count", "interaction": "impression", "scope": "item",
                  "bucket": "24h", "periods": [7, 30]})
    extra.append({"name": "emb", "type": "vector", "scope": "item", "source": "metadata.emb", "reduce": ["vector8"]})
    extra.append({"name": "lang", "type": "string", "scope": "item", "source": "metadata.lang",
                  "values": ["en", "fr", "de", "es", "it", "ja", "ko"]})
    cfg["features"] += extra
    m = cfg["models"]["xgboost"]
    m["features"] = m["features"] + ["day_item_click_count", "week_item_impressions", "global_item_click_count"] + \
        [f"x{i}" for i in range(20)] + ["emb", "lang"]
    return cfg


def c5_config(dim: int = 384) -> dict:
    """C5: the 24 Ranklens columns + one bi-encoder `field_match` column (cosine between the query embedding
    the host supplies as the request field `__embedding:title_match` and the stored item embedding)."""
    cfg = ranklens_config()
    cfg["features"].append({"name": "title_match", "type": "field_match", "itemField": "item.title", "rankingField": "ranking.query",
                            "method": {"type": "bi-encoder", "model": "metarank/all-MiniLM-L6-v2", "dim": dim}, "distance": "cosine"})
    cfg["models"]["xgboost"]["features"] = cfg["models"]["xgboost"]["features"] + ["title_match"]
    return cfg


def c5_embeddings(n_items: int, dim: int = 384, seed: int = SEED + 7, missing_frac: float = 0.05):
    """(kind, key, value) puts of the item embeddings: f32 values widened to f64 (Scalar.scala:24-32), one
    all-zero vector (cosine -> NaN, no epsilon) and a few items without an embedding"""
    rng = np.random.Generator(np.random.PCG64(seed))
    for i in range(n_items):
        if rng.random() < missing_frac:
            continue
        v = rng.normal(size=dim).astype(np.float32).astype(np.float64)
        if i == 7:
            v[:] = 0.0
        yield "double_list", f"item={i}/title_match", v


def c5_query(dim: int = 384, seed: int = 1) -> list:
    rng = np.random.Generator(np.random.PCG64(SEED + 1000 + seed))
    return [float(x) for x in rng.normal(size=dim).astype(np.float32)]


def _zipf_choice(rng, n, size, s=1.1):
    w = 1.0 / np.arange(1, n + 1) ** s
    return rng.choice(n, size=size, p=w / w.sum())


def generate_state(n_items=100_000, n_sessions=10_000, seed=SEED, c3=False):
    """Yields (kind, key, value) puts.  kind in double|string|string_list|double_list|counter|periodic|bounded_list."""
    rng = np.random.Generator(np.random.PCG64(seed))
    I = n_items
    no_meta = rng.random(I) < 0.05
    no_counter = rng.random(I) < 0.10
    pop = rng.integers(0, 100000, I)
    vote_avg = rng.integers(0, 10, I)
    vote_cnt = rng.integers(0, 1000, I)
    budget = rng.integers(0, 10**8, I)
    runtime = 50 + rng.integers(0, 120, I)
    release = rng.integers(10**9, 1_660_000_000, I)
    title_len = rng.integers(1, 7, I)
    n_gen = rng.integers(1, 4, I)
    actors = rng.integers(0, 5000, (I, 3))
    tags = _zipf_choice(rng, 1000, (I, 5))
    director = rng.integers(0, 2000, I)
    impr = np.minimum(rng.poisson(200 * (1.0 / (1 + _zipf_choice(rng, 50, I)))) + rng.integers(0, 50, I), 10**6)
    p = rng.beta(2, 20, I)
    clicks30 = rng.binomial(impr, p)
    impr7 = rng.binomial(impr, 0.3)
    clicks7 = np.minimum(rng.binomial(clicks30, 0.3), impr7)
    has_field_rate = rng.random(I) < 0.5
    g_click = np.zeros(2, dtype=np.int64)
    g_impr = np.zeros(2, dtype=np.int64)
    tag_cnt: dict = {}
    gen_cnt: dict = {}
    gen_perm = [rng.permutation(20) for _ in range(64)]
    for i in range(I):
        iid = str(i)
        k = f"item={iid}/"
        genres = [GENRES[j] for j in gen_perm[i % 64][:n_gen[i]]]
        tg = [f"tag{t}" for t in tags[i]]
        if not no_meta[i]:
            yield "double", k + "popularity", float(pop[i])
            yield "double", k + "vote_avg", float(vote_avg[i])
            yield "double", k + "vote_cnt", float(vote_cnt[i])
            yield "double", k + "budget", float(budget[i])
            yield "double", k + "release_date", float(release[i])
            yield "double", k + "runtime", float(runtime[i])
            yield "double", k + "title_length", float(title_len[i])
            yield "string_list", k + "genre", genres
            ac = [f"actor{a}" for a in actors[i]]
            for pref in ("profile_", "divers_"):
                yield "string_list", k + pref + "genres", genres
                yield "string_list", k + pref + "actors", ac
                yield "string_list", k + pref + "tags", tg
            yield "string_list", k + "profile_director", [f"dir{director[i]}"]
            yield "double", k + "divers_year", float(release[i])
            yield "double", k + "divers_popularity", float(pop[i])
            if has_field_rate[i]:
                yield "string", k + "ctr_tag_field", tg[0]
                yield "string", k + "ctr_genre_field", genres[0]
            if c3:
                for j in range(20):
                    yield "double", k + f"x{j}", float(np.float32(rng.normal() * (j + 1)))
                yield "double_list", k + "emb", rng.normal(size=8).astype(np.float32).astype(np.float64)
                yield "string_list", k + "lang", [["en", "fr", "de", "es", "it", "ja", "ko", "xx"][int(rng.integers(8))]]
        if not no_counter[i]:
            c = [int(clicks7[i]), int(clicks30[i])]
            m = [int(impr7[i]), int(impr[i])]
            yield "periodic", k + "ctr_click", c
            yield "periodic", k + "ctr_impression", m
            yield "counter", k + "global_item_click_count", int(clicks30[i])
            yield "periodic", k + "day_item_click_count", c
            if c3:
                yield "periodic", k + "week_item_impressions", m
            g_click += c
            g_impr += m
            if not no_meta[i] and has_field_rate[i]:
                for d, key in ((tag_cnt, tg[0]), (gen_cnt, genres[0])):
                    e = d.setdefault(key, [np.zeros(2, dtype=np.int64), np.zeros(2, dtype=np.int64)])
                    e[0] += c
                    e[1] += m
    yield "periodic", "global/ctr_click_norm", [int(x) for x in g_click]
    yield "periodic", "global/ctr_impression_norm", [int(x) for x in g_impr]
    for field, d in (("tag", tag_cnt), ("genre", gen_cnt)):
        for v, (c, m) in d.items():
            yield "periodic", f"field={field}:{v}/ctr_{field}_click", [int(x) for x in c]
            yield "periodic", f"field={field}:{v}/ctr_{field}_impression", [int(x) for x in m]
    for s in range(n_sessions):
        ln = int(rng.integers(0, 101))
        if ln:
            yield "bounded_list", f"session=s{s}/profile_interactions", [str(x) for x in rng.integers(0, I, ln)]
        if rng.random() < 0.7:
            yield "counter", f"session=s{s}/visitor_click_count", ln


def load_state(backend, puts):
    """backend: anything with put_double/put_string/... (HipRanker, tests' OracleBackend)."""
    fn = {"double": backend.put_double, "string": backend.put_string, "string_list": backend.put_string_list,
          "double_list": backend.put_double_list, "counter": backend.put_counter, "periodic": backend.put_periodic,
          "bounded_list": backend.put_bounded_list}
    n = 0
    for kind, key, value in puts:
        fn[kind](key, value)
        n += 1
    return n


def generate_requests(n_requests, n_items_per_request=100, catalogue=100_000, n_sessions=10_000, seed=SEED + 1,
                      unknown_frac=0.01):
    """RankingEvent dicts: distinct item ids drawn uniformly, a uniformly drawn session, ts = TS."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for r in range(n_requests):
        ids = rng.choice(catalogue, size=n_items_per_request, replace=False) if n_items_per_request <= catalogue \
            else rng.integers(0, catalogue, n_items_per_request)
        items = [{"id": (str(x) if rng.random() >= unknown_frac else f"unknown{x}")} for x in ids]
        sess = int(rng.integers(0, int(n_sessions * 1.05)))  # a few sessions have no state
        out.append({"id": f"req{r}", "timestamp": TS, "user": f"u{sess}", "session": f"s{sess}", "fields": [], "items": items})
    return out


def column_quantiles(matrix: np.ndarray, n=49):
    """per-column split candidates for synthetic forests (empirical quantiles of an assembled matrix)."""
    out = []
    for j in range(matrix.shape[1]):
        col = matrix[:, j]
        col = col[np.isfinite(col)]
        out.append(np.unique(np.quantile(col, np.linspace(0.02, 0.98, n))) if len(col) else np.array([0.0, 0.5]))
    return out
