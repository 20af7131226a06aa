"""One suite, N backends (the reference's own pattern): the CPU oracle and the HIP
product behind the same tiny interface, so every known-answer test runs against both."""
from __future__ import annotations

import numpy as np
import pytest

from metarank_amd.request import Request


class OracleBackend:
    name = "oracle"

    def __init__(self, config: dict, model: str):
        from oracle.assembly import OraclePlan, OracleStore

        from oracle.writes import WriteState

        self.store = OracleStore()
        self.plan = OraclePlan(config, model)
        self.dim = self.plan.dim
        self.forest = None
        self.writes = WriteState(config)

    # raw Writes (FeatureValueFlow.commitWrite + makeValue): the oracle derives the FeatureValue and puts it
    def increment_periodic(self, k, ts, inc=1): self.store.put_periodic(k, self.writes.increment_periodic(k, ts, inc))
    def increment(self, k, inc=1): self.store.put_counter(k, self.writes.increment(k, inc))
    def append(self, k, v, ts): self.store.put_bounded_list(k, self.writes.append(k, v, ts))

    # KVStore.put
    def put_double(self, k, v): self.store.put_double(k, v)
    def put_bool(self, k, v): self.store.put_bool(k, v)
    def put_string(self, k, v): self.store.put_string(k, v)
    def put_string_list(self, k, v): self.store.put_string_list(k, v)
    def put_double_list(self, k, v): self.store.put_double_list(k, v)
    def put_counter(self, k, v): self.store.put_counter(k, v)
    def put_periodic(self, k, v): self.store.put_periodic(k, v)
    def put_bounded_list(self, k, v): self.store.put_bounded_list(k, v)
    def delete(self, k): self.store.delete(k)

    def load_model(self, blob: bytes, backend: int):
        from oracle.forest import OracleForest

        self.forest = OracleForest.from_lightgbm_text(blob) if backend == 0 else OracleForest.from_xgboost(blob)

    def matrix(self, event: dict) -> np.ndarray:
        return self.plan.assemble(self.store, Request(event))

    def rerank(self, event: dict):
        """-> (matrix, scores, order)"""
        from oracle.assembly import sort_order

        m = self.matrix(event)
        scores = self.forest.predict(m) if self.forest is not None else np.zeros(len(m))
        return m, scores, sort_order(scores)

    def expect_throws(self):
        from oracle.assembly import ReferenceThrows

        return pytest.raises(ReferenceThrows)

    def close(self):
        pass


class HipBackend:
    name = "hip"

    def __init__(self, config: dict, model: str, ctx=None):
        import metarank_amd as M
        from metarank_amd.ranker import HipRanker

        self.M = M
        self.ctx_owned = ctx is None
        self.ctx = ctx or M.Context(0)
        self.ranker = HipRanker(config, self.ctx)
        self.model_name = model
        self.dim = self.ranker.dim(model)
        self.booster = None

    def put_double(self, k, v): self.ranker.put_double(k, v)
    def put_bool(self, k, v): self.ranker.put_bool(k, v)
    def put_string(self, k, v): self.ranker.put_string(k, v)
    def put_string_list(self, k, v): self.ranker.put_string_list(k, v)
    def put_double_list(self, k, v): self.ranker.put_double_list(k, v)
    def put_counter(self, k, v): self.ranker.put_counter(k, v)
    def put_periodic(self, k, v): self.ranker.put_periodic(k, v)
    def put_bounded_list(self, k, v): self.ranker.put_bounded_list(k, v)
    def delete(self, k): self.ranker.delete(k)
    def increment_periodic(self, k, ts, inc=1): self.ranker.increment_periodic(k, ts, inc)
    def increment(self, k, inc=1): self.ranker.increment(k, inc)
    def append(self, k, v, ts): self.ranker.append(k, v, ts)

    def load_model(self, blob: bytes, backend: int):
        self.booster = self.M.HipBooster(blob, backend, self.ctx)

    def matrix(self, event: dict) -> np.ndarray:
        return self.rerank(event)[0]

    def rerank(self, event: dict):
        return self.ranker.rerank(self.model_name, event, self.booster, explain=True)

    def expect_throws(self):
        return pytest.raises(self.M.MrkError)

    def close(self):
        self.ranker.close()
        if self.ctx_owned:
            self.ctx.close()


def make_backend(kind: str, config: dict, model: str):
    return OracleBackend(config, model) if kind == "oracle" else HipBackend(config, model)


BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def single_feature_config(feature: dict) -> dict:
    """FeatureTest.process: one schema, model "random" with that single feature (T/feature/FeatureTest.scala:17-23)."""
    return {"features": [feature], "models": {"random": {"type": "lambdamart", "features": [feature["name"]]}}}


def ranking_event(items, **kw) -> dict:
    """TestRankingEvent(items): user u1, session s1, no fields (T/util/TestRankingEvent.scala:11-19)."""
    ev = {"id": "r-test", "timestamp": 1661345221008, "user": "u1", "session": "s1", "fields": [],
          "items": [it if isinstance(it, dict) else {"id": it} for it in items]}
    ev.update(kw)
    return ev
