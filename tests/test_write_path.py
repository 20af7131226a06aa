"""The write side that feeds /rank: raw Writes -> derived FeatureValues.

(1) the oracle's restatement of the reference's raw-state primitives against the reference's own
    known-answer suites; (2) the same answers through the product (HIP bucket rings + window-sum
    kernels behind mrk_store_increment_periodic, host-side bounded lists / counters) read back
    through the rank path; (3) randomised streams, oracle vs product."""
import datetime

import numpy as np
import pytest

from backends import BACKENDS, make_backend, ranking_event, single_feature_config
from oracle.writes import BoundedList, PeriodicCounter, start_of_period

DAY, HOUR = 86_400_000, 3_600_000
NOW = int(datetime.datetime(2021, 6, 1, 0, 0, 1, tzinfo=datetime.timezone.utc).timestamp() * 1000)  # FeatureSuite.now


# ---- T/fstore/PeriodicCounterSuite.scala:22-144 : config 1.day, PeriodRange(0,0), PeriodRange(7,0) ----
SUITE_CASES = {
    "once": ([NOW], 1, 1),
    "intra_day_burst": ([NOW - 10 * HOUR + k * HOUR for k in range(1, 11)], 1, 10),
    "once_a_day": ([NOW - 10 * DAY + k * DAY for k in range(1, 11)], 1, 8),
    "once_a_week": ([NOW - 70 * DAY + 7 * k * DAY for k in range(1, 11)], 1, 2),
}


@pytest.mark.parametrize("case", list(SUITE_CASES))
def test_oracle_periodic_counter_suite(case):
    stamps, today, week = SUITE_CASES[case]
    c = PeriodicCounter(DAY, [(0, 0), (7, 0)])
    for ts in stamps:
        c.put(ts, 1)
    day0 = start_of_period(NOW, DAY)
    assert c.values() == [(day0, day0 + DAY, 1, today), (day0 - 7 * DAY, day0 + DAY, 8, week)]
    assert PeriodicCounter(DAY, [(0, 0)]).values() is None  # "be empty"


def test_oracle_bounded_list_suite():  # T/fstore/BoundedListSuite.scala:24-68 : count 10, duration 5 h
    l = BoundedList(10, 5 * HOUR)
    l.put("foo", NOW)
    assert l.values() == ["foo"]
    l.put("bar", NOW + 1000)
    assert l.values() == ["bar", "foo"]
    l = BoundedList(10, 5 * HOUR)
    for i in range(10):
        l.put(str(i), NOW + i)
    assert len(l.values()) == 10
    l.put("x", NOW + 11)
    assert len(l.values()) == 10 and l.values()[0] == "x"  # bounded by element count
    l = BoundedList(10, 5 * HOUR)
    for i in reversed(range(10)):
        l.put(str(i), NOW - i * HOUR)
    assert all(ts >= NOW - 5 * HOUR for ts, _ in l.items)  # bounded by time


WINDOW = {"name": "cnt", "type": "window_count", "interaction": "click", "scope": "item", "bucket": "1d", "periods": [0, 7]}


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("case", list(SUITE_CASES))
def test_periodic_counter_suite_through_the_rank_path(kind, case):
    stamps, today, week = SUITE_CASES[case]
    b = make_backend(kind, single_feature_config(WINDOW), "random")
    try:
        for ts in stamps:
            b.increment_periodic("item=p1/cnt", ts, 1)
        m = b.matrix(ranking_event(["p1", "p2"]))
        assert m[0].tolist() == [float(today), float(week)]
        assert np.isnan(m[1]).all()  # never incremented: missing -> NaN x dim
    finally:
        b.close()


@pytest.mark.parametrize("kind", BACKENDS)
def test_window_count_three_clicks(kind):  # T/feature/WindowInteractionCountFeatureTest.scala:46-57
    b = make_backend(kind, single_feature_config(dict(WINDOW, bucket="24h", periods=[1])), "random")
    try:
        for _ in range(3):
            b.increment_periodic("item=p1/cnt", 1661345221008, 1)
        assert b.matrix(ranking_event(["p1"]))[0].tolist() == [3.0]
    finally:
        b.close()


RATE = {"name": "ctr", "type": "rate", "top": "click", "bottom": "impression", "bucket": "24h", "periods": [7, 14]}


@pytest.mark.parametrize("kind", BACKENDS)
def test_rate_from_raw_increments(kind):  # T/feature/RateFeatureTest.scala:61-74: 1 click / 4 impressions -> [0.25, 0.25]
    b = make_backend(kind, single_feature_config(RATE), "random")
    try:
        ts = 1661345221008
        b.increment_periodic("item=p1/ctr_click", ts, 1)
        for k in range(4):
            b.increment_periodic("item=p1/ctr_impression", ts + k, 1)
        assert b.matrix(ranking_event(["p1"]))[0].tolist() == [0.25, 0.25]
    finally:
        b.close()


IW = {"name": "profile", "type": "interacted_with", "interaction": "click", "field": ["item.genres"], "scope": "session",
      "count": 3, "duration": "1h"}


@pytest.mark.parametrize("kind", BACKENDS)
def test_bounded_list_and_counter_writes(kind):
    cfg = {"features": [IW, {"name": "clicks", "type": "interaction_count", "interaction": "click", "scope": "item"}],
           "models": {"random": {"type": "lambdamart", "features": ["profile", "clicks"]}}}
    b = make_backend(kind, cfg, "random")
    try:
        for it, g in (("a", ["x"]), ("b", ["x", "y"]), ("c", ["y"]), ("d", ["z"])):
            b.put_string_list(f"item={it}/profile_genres", g)
        t0 = 1661345221008
        b.append("session=s1/profile_interactions", "a", t0)                 # [a]
        b.append("session=s1/profile_interactions", "b", t0 + 60_000)        # [b, a]
        m = b.matrix(ranking_event(["a", "b", "c", "d"]))
        assert m[:, 0].tolist() == [2.0, 3.0, 1.0, 0.0]   # histogram {x: 2, y: 1}
        b.append("session=s1/profile_interactions", "c", t0 + 2 * HOUR)      # a and b are older than 1 h: [c]
        assert b.matrix(ranking_event(["a", "b", "c", "d"]))[:, 0].tolist() == [0.0, 1.0, 1.0, 0.0]
        for k, it in enumerate("dddd"):
            b.append("session=s1/profile_interactions", it, t0 + 2 * HOUR + k)  # count 3: [d, d, d]
        assert b.matrix(ranking_event(["a", "b", "c", "d"]))[:, 0].tolist() == [0.0, 0.0, 0.0, 3.0]
        b.increment("item=a/clicks", 2)
        b.increment("item=a/clicks", 3)
        assert b.matrix(ranking_event(["a", "b"]))[:, 1].tolist() == [5.0, 0.0]
    finally:
        b.close()


@pytest.mark.gpu
def test_random_increment_streams_match_the_oracle():
    """Out-of-order timestamps, many keys, several flushes, puts to other columns in between (their row uploads
    must not clobber the ring-fed cells), table growth: matrices equal the oracle's after every flush."""
    cfg = {"features": [dict(WINDOW, periods=[0, 1, 7, 30]), RATE, {"name": "pop", "type": "number", "scope": "item", "source": "item.pop"}],
           "models": {"random": {"type": "lambdamart", "features": ["cnt", "ctr", "pop"]}}}
    hip, orc = make_backend("hip", cfg, "random"), make_backend("oracle", cfg, "random")
    try:
        rng = np.random.default_rng(5)
        t0 = 1661345221008
        n_items = 40
        for rnd in range(6):
            n_items += 25  # new slots every round: the device tables (and the rings) grow
            for _ in range(1500):
                it = int(rng.integers(n_items))
                state = ["cnt", "ctr_click", "ctr_impression"][int(rng.integers(3))]
                ts = t0 + int(rng.integers(-45 * DAY, 2 * DAY)) + rnd * 3 * DAY
                inc = int(rng.integers(1, 4))
                for b in (hip, orc):
                    b.increment_periodic(f"item=i{it}/{state}", ts, inc)
            for it in rng.integers(n_items, size=30):
                for b in (hip, orc):
                    b.put_double(f"item=i{int(it)}/pop", float(rnd))
            ev = ranking_event([f"i{k}" for k in rng.permutation(n_items)[:60]] + ["nobody"])
            a, e = hip.matrix(ev), orc.matrix(ev)
            assert ((a == e) | (np.isnan(a) & np.isnan(e))).all(), rnd
    finally:
        hip.close()
        orc.close()
