"""Known-answer tests transcribed from the reference's own test suite, run
against BOTH the CPU oracle (not gpu) and the HIP product through the C ABI (gpu).

The reference tests push events through its write path and then read; the write path is out of
scope here, so each test puts the FeatureValues that write path produces (cited per test) and
asserts the reference's expected read-side values.  T = /root/reference/src/test/scala/ai/metarank.
"""
import math

import numpy as np
import pytest

from backends import BACKENDS, make_backend, ranking_event, single_feature_config

NAN = float("nan")


@pytest.fixture(params=BACKENDS)
def mk(request):
    made = []

    def factory(config, model="random"):
        b = make_backend(request.param, config, model)
        made.append(b)
        return b

    yield factory
    for b in made:
        b.close()


def eq(got, exp):
    got = np.asarray(got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (got.shape, exp.shape)
    same = (got == exp) | (np.isnan(got) & np.isnan(exp))
    assert same.all(), f"\n got {got.tolist()}\n exp {exp.tolist()}"


# ---- T/flow/ClickthroughQueryTest.scala:103-159 : matrix layout, category + vector columns ----------
def test_matrix_layout_category_and_vector_columns(mk):
    cfg = {
        "features": [
            {"name": "price", "type": "number", "scope": "item", "source": "item.price"},
            {"name": "category", "type": "string", "scope": "item", "source": "item.category", "encode": "index",
             "values": ["socks", "shirts"]},
            {"name": "ctr", "type": "rate", "top": "click", "bottom": "impression", "bucket": "24h", "periods": [7, 30]},
            {"name": "clicked_category", "type": "interacted_with", "interaction": "click", "field": "item.category",
             "scope": "session"},
        ],
        "models": {"xgboost": {"type": "lambdamart", "features": ["price", "category", "ctr", "clicked_category"]}},
    }
    b = mk(cfg, "xgboost")
    assert b.dim == 5
    # state that yields the MValues listed in the reference test
    for item, price, cat, ctr, clicked in [("p1", 10.0, "socks", (2, 10, 1, 10), 1), ("p2", 5.0, "shirts", (1, 10, 1, 20), 0),
                                           ("p3", 3.0, "socks", (2, 10, 2, 10), 1)]:
        b.put_double(f"item={item}/price", price)
        b.put_string_list(f"item={item}/category", [cat])
        b.put_periodic(f"item={item}/ctr_click", [ctr[0], ctr[2]])
        b.put_periodic(f"item={item}/ctr_impression", [ctr[1], ctr[3]])
        b.put_string_list(f"item={item}/clicked_category_category", [cat])
    b.put_string_list("item=p0/clicked_category_category", ["socks"])
    b.put_bounded_list("session=s1/clicked_category_interactions", ["p0"])
    m = b.matrix(ranking_event(["p1", "p2", "p3"]))
    eq(m, [[10.0, 1.0, 0.2, 0.1, 1.0], [5.0, 2.0, 0.1, 0.05, 0.0], [3.0, 1.0, 0.2, 0.2, 1.0]])


# ---- T/feature/RateFeatureTest.scala:61-74 ------------------------------------------------------------
RATE = {"name": "ctr", "type": "rate", "top": "click", "bottom": "impression", "bucket": "24h", "periods": [7, 14], "refresh": "0s"}


def test_rate_plain(mk):
    b = mk(single_feature_config(RATE))
    b.put_periodic("item=p1/ctr_click", [1, 1])
    b.put_periodic("item=p1/ctr_impression", [4, 4])
    eq(b.matrix(ranking_event(["p1"])), [[0.25, 0.25]])


def test_rate_missing_and_wrong_length_and_division_by_zero(mk):
    b = mk(single_feature_config(RATE))
    b.put_periodic("item=p1/ctr_click", [1, 1])  # no impressions at all -> missing
    b.put_periodic("item=p2/ctr_click", [1])  # wrong length
    b.put_periodic("item=p2/ctr_impression", [4, 4])
    b.put_periodic("item=p3/ctr_click", [0, 3])  # Long / Double: 0/0 = NaN, 3/0 = +Inf
    b.put_periodic("item=p3/ctr_impression", [0, 0])
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"])), [[NAN, NAN], [NAN, NAN], [NAN, math.inf], [NAN, NAN]])


# ---- T/feature/NormRateFeatureTest.scala:64-81 : Long / Long inside the normalisation -------------------
def test_rate_normalized_long_division(mk):
    b = mk(single_feature_config(dict(RATE, normalize={"weight": 10})))
    b.put_periodic("item=p1/ctr_click", [1, 1])
    b.put_periodic("item=p1/ctr_impression", [3, 3])
    b.put_periodic("global/ctr_click_norm", [10, 10])
    b.put_periodic("global/ctr_impression_norm", [93, 93])
    eq(b.matrix(ranking_event(["p1"])), [[0.11827956989247312, 0.11827956989247312]])


def test_rate_normalized_long_division_at_the_edges_of_the_fast_path(mk):
    """(Long / Long).toDouble of the global counters: the device takes the quotient of the doubles, truncated, where that is
    exact (0 <= a, 0 < b, both below 2^52: rank_device.hpp long_div_to_double) and the integer division elsewhere - the cases
    sit on both sides of that boundary, on quotients one below a multiple, on negative operands and on Long.MinValue / -1."""
    def jdiv(a, b):  # Scala Long / Long: truncates toward zero, wraps at Long.MinValue / -1
        q = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)
        return (q + 2 ** 63) % 2 ** 64 - 2 ** 63

    big = 2 ** 52
    cases = [(93, 10), (big - 1, 1), (big - 1, big - 1), (big - 2, big - 1), (big - 1, 3), (big, 3), (big - 1, big), (big + 1, 7), (2 ** 53 + 1, 3),
             (3 * (2 ** 50) - 1, 3), ((2 ** 26 + 1) * (2 ** 25 - 1) - 1, 2 ** 25 - 1), (2 ** 62 + 12345, 2 ** 31 + 7), (-93, 10), (93, -10), (-(2 ** 63), -1),
             (-(2 ** 63), 1), (2 ** 63 - 1, 2), (0, 5), (4, 5), (5, 5)]
    for gimp, gclick in cases:
        b = mk(single_feature_config(dict(RATE, normalize={"weight": 10})))
        b.put_periodic("item=p1/ctr_click", [1, 2])
        b.put_periodic("item=p1/ctr_impression", [3, 5])
        b.put_periodic("global/ctr_click_norm", [gclick, 10])
        b.put_periodic("global/ctr_impression_norm", [gimp, 93])
        ratio = float(jdiv(gimp, gclick))
        exp = (10.0 + 1.0) / (10.0 * ratio + 3.0)
        eq(b.matrix(ranking_event(["p1"])), [[exp, (10.0 + 2.0) / (10.0 * 9.0 + 5.0)]])


def test_rate_normalized_zero_global_clicks_throws(mk):
    b = mk(single_feature_config(dict(RATE, normalize={"weight": 10})))
    b.put_periodic("item=p1/ctr_click", [1, 1])
    b.put_periodic("item=p1/ctr_impression", [3, 3])
    b.put_periodic("global/ctr_click_norm", [0, 10])
    b.put_periodic("global/ctr_impression_norm", [93, 93])
    with b.expect_throws():  # java.lang.ArithmeticException: / by zero  (RateFeature.scala:346-348)
        b.matrix(ranking_event(["p1"]))
    # an item without state never reaches the division
    eq(b.matrix(ranking_event(["p9"])), [[NAN, NAN]])


# ---- T/feature/ScopedRateFeatureTest.scala:78-124 : item-field scope ----------------------------------------
def test_rate_item_field_scope(mk):
    b = mk(single_feature_config(dict(RATE, scope="item.color")))
    for p, c in [("p1", "red"), ("p2", "red"), ("p3", "red"), ("p4", "green")]:
        b.put_string(f"item={p}/ctr_field", c)
    b.put_periodic("field=color:red/ctr_click", [1, 1])
    b.put_periodic("field=color:red/ctr_impression", [4, 4])
    b.put_periodic("field=color:green/ctr_click", [1, 1])
    b.put_periodic("field=color:green/ctr_impression", [1, 1])
    eq(b.matrix(ranking_event(["p1"])), [[0.25, 0.25]])
    eq(b.matrix(ranking_event(["p4", "p5", "p2"])), [[1.0, 1.0], [NAN, NAN], [0.25, 0.25]])


# ---- T/feature/RankFieldScopedRateFeatureTest.scala:48-66 : ranking-field scope --------------------------
def test_rate_ranking_field_scope(mk):
    b = mk(single_feature_config(dict(RATE, scope="ranking.query")))
    b.put_periodic("irf=query:test:p1/ctr_click", [1, 1])
    b.put_periodic("irf=query:test:p1/ctr_impression", [2, 2])
    b.put_periodic("irf=query:test:p2/ctr_click", [1, 1])
    b.put_periodic("irf=query:test:p2/ctr_impression", [2, 2])
    ev = ranking_event(["p1"], fields=[{"name": "query", "value": "test"}])
    eq(b.matrix(ev), [[0.5, 0.5]])
    eq(b.matrix(ranking_event(["p1"], fields=[{"name": "query", "value": "other"}])), [[NAN, NAN]])
    eq(b.matrix(ranking_event(["p1"])), [[NAN, NAN]])


# ---- T/feature/WindowInteractionCountFeatureTest.scala:46-57, T/util/FeatureMappingTest.scala:17-43 ---------
def test_window_count(mk):
    b = mk(single_feature_config({"name": "cnt", "type": "window_count", "interaction": "click", "scope": "item",
                                  "bucket": "24h", "periods": [1]}))
    assert b.dim == 1  # a 1-period window_count is still a VectorFeature(name, 1)
    b.put_periodic("item=p1/cnt", [3])
    b.put_periodic("item=p2/cnt", [3, 4])  # length != dim
    eq(b.matrix(ranking_event(["p1", "p2", "p3"])), [[3.0], [NAN], [NAN]])


# ---- T/feature/InteractionCountTest.scala:50-58 ------------------------------------------------------------
def test_interaction_count(mk):
    b = mk(single_feature_config({"name": "cnt", "type": "interaction_count", "interaction": "click", "scope": "item"}))
    b.put_counter("item=p1/cnt", 3)
    eq(b.matrix(ranking_event(["p1", "p2"])), [[3.0], [0.0]])  # missing => 0.0, not NaN


def test_interaction_count_session_scope(mk):
    b = mk(single_feature_config({"name": "cnt", "type": "interaction_count", "interaction": "click", "scope": "session"}))
    b.put_counter("session=s1/cnt", 7)
    eq(b.matrix(ranking_event(["p1", "p2"])), [[7.0], [7.0]])
    eq(b.matrix(ranking_event(["p1"], session=None)), [[0.0]])


# ---- T/feature/InteractedWithFeatureTest.scala:105-144 ---------------------------------------------------------
IW = {"name": "seen", "type": "interacted_with", "interaction": "impression", "field": "item.color", "scope": "session",
      "count": 10, "duration": "24h"}


def test_interacted_with_single_field(mk):
    b = mk(single_feature_config(IW))
    b.put_string_list("item=p1/seen_color", ["red"])
    b.put_string_list("item=p2/seen_color", ["green"])
    b.put_bounded_list("session=s1/seen_interactions", ["p2", "p1"])
    eq(b.matrix(ranking_event(["p1", "p2", "p3"])), [[1.0], [1.0], [0.0]])


def test_interacted_with_two_fields(mk):
    b = mk(single_feature_config(dict(IW, field=["item.color", "item.tags"])))
    b.put_string_list("item=p1/seen_color", ["red"])
    b.put_string_list("item=p2/seen_color", ["green"])
    b.put_bounded_list("session=s1/seen_interactions", ["p2", "p1"])
    eq(b.matrix(ranking_event(["p1", "p2", "p3"])), [[1.0, 0.0], [1.0, 0.0], [0.0, 0.0]])


def test_interacted_with_list_fields_and_repeats(mk):
    b = mk(single_feature_config(IW))
    b.put_string_list("item=p1/seen_color", ["red", "green", "blue"])
    b.put_string_list("item=p2/seen_color", ["brown", "red", "green"])
    b.put_string_list("item=p3/seen_color", ["red", "red"])  # duplicates in the candidate count twice
    b.put_bounded_list("session=s1/seen_interactions", ["p1", "p1", "p9"])  # repeated interaction counts twice
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"])), [[6.0], [4.0], [4.0], [0.0]])
    eq(b.matrix(ranking_event(["p1"], session="nobody")), [[0.0]])
    eq(b.matrix(ranking_event(["p1"], session=None)), [[0.0]])


def test_interacted_with_five_fields_come_in_scala_map_order(mk):
    """more than 4 fields: `fields` is a HashMap and the columns follow ITS iteration order (InteractedWithFeature.scala:56-65,
    152-162) - for the keys a..e that is e, a, b, c, d (what a Scala 2.13 REPL prints for Map("a"->1,...,"e"->5);
    tests/test_host_planning_cpu.py).  Field k of the interacted item holds k + 1 equal tokens: the count is (k + 1)^2."""
    b = mk(single_feature_config(dict(IW, field=[f"item.{c}" for c in "abcde"])))
    for k, c in enumerate("abcde"):
        b.put_string_list(f"item=p1/seen_{c}", ["t"] * (k + 1))
    b.put_bounded_list("session=s1/seen_interactions", ["p1"])
    eq(b.matrix(ranking_event(["p1", "p2"])), [[25.0, 1.0, 4.0, 9.0, 16.0], [0.0] * 5])


# ---- T/feature/DiversityFeatureTest.scala:16-101 ------------------------------------------------------------------
def test_diversity_numbers(mk):
    b = mk(single_feature_config({"name": "divnum", "type": "diversity", "source": "item.price", "top": 2147483647}))
    for p, v in zip(["p1", "p2", "p3", "p4", "p5"], [10.0, 20.0, 40.0, 15.0, 5.0]):
        b.put_double(f"item={p}/divnum", v)
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4", "p5"])), [[-5.0], [5.0], [25.0], [0.0], [-10.0]])


def test_diversity_top_n_numbers(mk):
    b = mk(single_feature_config({"name": "divnum", "type": "diversity", "source": "item.price", "top": 3}))
    for p, v in zip(["p1", "p2", "p3", "p4", "p5"], [10.0, 20.0, 30.0, 5.0, 1.0]):
        b.put_double(f"item={p}/divnum", v)
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4", "p5"])), [[-10.0], [0.0], [10.0], [-15.0], [-19.0]])


def test_diversity_even_count_interpolates(mk):
    b = mk(single_feature_config({"name": "d", "type": "diversity", "source": "item.price"}))
    for p, v in zip(["p1", "p2", "p3", "p4"], [1.0, 2.0, 4.0, 8.0]):
        b.put_double(f"item={p}/d", v)
    # commons-math LEGACY percentile: pos = 0.5 * (4 + 1) = 2.5 -> 2 + 0.5 * (4 - 2) = 3
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4", "p5"])), [[-2.0], [-1.0], [1.0], [5.0], [NAN]])


def test_diversity_strings(mk):
    b = mk(single_feature_config({"name": "divstr", "type": "diversity", "source": "item.cat", "top": 2147483647}))
    for p, v in zip(["p1", "p2", "p3", "p4", "p5"], ["a", "b", "c", "a", "b"]):
        b.put_string(f"item={p}/divstr", v)
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4", "p5"])), [[0.4], [0.4], [0.2], [0.4], [0.4]])


def test_diversity_string_lists(mk):
    b = mk(single_feature_config({"name": "divstrl", "type": "diversity", "source": "item.cat", "top": 2147483647}))
    for p, v in zip(["p1", "p2", "p3", "p4"], [["a"], ["b", "c"], ["a", "b", "c"], ["a", "b", "c", "d"]]):
        b.put_string_list(f"item={p}/divstrl", v)
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"])), [[0.3], [0.6], [0.9], [1.0]])


def test_diversity_nobody_has_state_and_default_top_20(mk):
    b = mk(single_feature_config({"name": "d", "type": "diversity", "source": "item.cat"}))
    eq(b.matrix(ranking_event(["p1", "p2"])), [[0.0], [0.0]])  # emptyResponse
    items = [f"p{i}" for i in range(30)]
    for i, p in enumerate(items):
        b.put_string_list(f"item={p}/d", ["x"] if i < 25 else ["y"])
    b.delete("item=p3/d")
    m = b.matrix(ranking_event(items))
    # histogram over the first 20 PRESENT items (p0..p20 without p3): x=20, sum=20
    exp = [[1.0] if i < 25 else [0.0] for i in range(30)]
    exp[3] = [NAN]
    eq(m, exp)


def test_diversity_type_decided_by_first_present_item(mk):
    b = mk(single_feature_config({"name": "d", "type": "diversity", "source": "item.cat"}))
    b.put_double("item=p2/d", 10.0)
    b.put_string("item=p3/d", "a")  # other type: silently dropped
    b.put_double("item=p4/d", 20.0)
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"])), [[NAN], [-5.0], [NAN], [5.0]])


# ---- T/feature/StringFeatureTest.scala:101-171, T/util/OneHotEncoderTest.scala -----------------------------------
COLOR = {"name": "color", "type": "string", "scope": "item", "source": "item.color", "values": ["red", "green", "blue"]}


def test_string_onehot_item(mk):
    b = mk(single_feature_config(COLOR))
    b.put_string_list("item=p1/color", ["green"])
    b.put_string_list("item=p2/color", ["blue", "red", "pink"])
    eq(b.matrix(ranking_event(["p1", "p2", "p3"])), [[0.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 0.0, 0.0]])


def test_string_ranking_field(mk):
    b = mk(single_feature_config(dict(COLOR, source="ranking.color")))
    eq(b.matrix(ranking_event(["p1"], fields=[{"name": "color", "value": "red"}])), [[1.0, 0.0, 0.0]])
    eq(b.matrix(ranking_event(["p1", "p2"])), [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])


def test_string_session_scope(mk):
    b = mk(single_feature_config({"name": "country", "type": "string", "scope": "session", "source": "interaction:click.country",
                                  "values": ["US", "EU"]}))
    b.put_string_list("session=s1/country", ["EU"])
    eq(b.matrix(ranking_event(["p1"])), [[0.0, 1.0]])


def test_string_override_from_rank_event(mk):
    b = mk(single_feature_config(COLOR))
    b.put_string_list("item=p1/color", ["green"])
    ev = ranking_event([{"id": "p1", "fields": [{"name": "color", "value": "red"}]}, {"id": "p1"}])
    eq(b.matrix(ev), [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])


def test_string_index_encoding(mk):
    b = mk(single_feature_config(dict(COLOR, encode="index")))
    assert b.dim == 1
    b.put_string_list("item=p1/color", ["green", "red"])  # first element decides
    b.put_string_list("item=p2/color", ["pink"])
    b.put_string_list("item=p3/color", ["blue"])
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"])), [[2.0], [0.0], [3.0], [0.0]])  # unknown / missing => 0


# ---- T/feature/NumberFeatureTest.scala:89-124 -----------------------------------------------------------------------
POP = {"name": "popularity", "type": "number", "scope": "item", "source": "item.popularity"}


def test_number(mk):
    b = mk(single_feature_config(POP))
    b.put_double("item=p1/popularity", 100.0)
    b.put_bool("item=p2/popularity", True)  # wrong scalar type => missing
    eq(b.matrix(ranking_event(["p1", "p2", "p3"])), [[100.0], [NAN], [NAN]])


def test_number_field_override(mk):
    b = mk(single_feature_config(POP))
    b.put_double("item=p1/popularity", 1.0)
    ev = ranking_event([{"id": "p1", "fields": [{"name": "popularity", "value": 100}]}, {"id": "p1"}])
    eq(b.matrix(ev), [[100.0], [1.0]])


def test_number_ranking_scope(mk):
    b = mk(single_feature_config({"name": "weather_temp", "type": "number", "scope": "ranking", "source": "ranking.temp"}))
    eq(b.matrix(ranking_event(["p1", "p2"], fields=[{"name": "temp", "value": 10}])), [[10.0], [10.0]])
    eq(b.matrix(ranking_event(["p1"])), [[NAN]])


def test_number_user_scope(mk):
    b = mk(single_feature_config({"name": "user_age", "type": "number", "scope": "user", "source": "user.age"}))
    b.put_double("user=u1/user_age", 33.0)
    eq(b.matrix(ranking_event(["p1", "p2"])), [[33.0], [33.0]])
    eq(b.matrix(ranking_event(["p1"], user="u2")), [[NAN]])
    eq(b.matrix(ranking_event(["p1"], user=None)), [[NAN]])


def test_boolean(mk):
    b = mk(single_feature_config({"name": "avail", "type": "boolean", "scope": "item", "field": "item.availability"}))
    b.put_bool("item=p1/avail", True)
    b.put_bool("item=p2/avail", False)
    b.put_double("item=p3/avail", 1.0)
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"])), [[1.0], [0.0], [NAN], [NAN]])


def test_word_count(mk):
    b = mk(single_feature_config({"name": "title_words", "type": "word_count", "source": "item.title", "scope": "item"}))
    b.put_double("item=p1/title_words", 2.0)
    eq(b.matrix(ranking_event(["p1", "p2"])), [[2.0], [NAN]])
    b2 = mk(single_feature_config({"name": "q_words", "type": "word_count", "source": "ranking.query", "scope": "ranking"}))
    eq(b2.matrix(ranking_event(["p1", "p2"], fields=[{"name": "query", "value": "red  cotton socks "}])), [[3.0], [3.0]])
    eq(b2.matrix(ranking_event(["p1"], fields=[{"name": "query", "value": " x"}])), [[2.0]])  # leading empty token is kept
    eq(b2.matrix(ranking_event(["p1"])), [[NAN]])


# ---- T/feature/NumVectorFeatureTest.scala (read side) ------------------------------------------------------------------
def test_vector(mk):
    b = mk(single_feature_config({"name": "v", "type": "vector", "source": "item.v", "scope": "item"}))
    assert b.dim == 4  # default reducers [min, max, size, avg]
    b.put_double_list("item=p1/v", [1.0, 3.0, 2.0, 2.0])
    eq(b.matrix(ranking_event(["p1", "p2"])), [[1.0, 3.0, 2.0, 2.0], [NAN] * 4])
    b2 = mk(single_feature_config({"name": "v", "type": "vector", "source": "item.v", "scope": "item", "reduce": ["vector3", "sum"]}))
    assert b2.dim == 4


# ---- T/feature/ItemAgeFeatureTest.scala:101-122 ----------------------------------------------------------------------
def test_item_age(mk):
    b = mk(single_feature_config({"name": "itemage", "type": "item_age", "source": "item.updated_at"}))
    updated_at = 1646085600  # 2022-03-01T00:00+02:00
    now_ms = 1648418400000  # 2022-03-28T00:00+02:00
    b.put_double("item=p1/itemage", float(updated_at))
    b.put_double("item=p2/itemage", updated_at + 0.9996)  # round(v*1000) then whole seconds
    b.put_double("item=p3/itemage", float(now_ms // 1000 + 100))  # in the future: |delta|
    eq(b.matrix(ranking_event(["p1", "p2", "p3", "p4"], timestamp=now_ms)), [[2332800.0], [2332799.0], [100.0], [NAN]])


# ---- T/feature/LocalDateTimeFeatureTest.scala --------------------------------------------------------------------------
@pytest.mark.parametrize("mapper,expected", [("time_of_day", 12.0), ("day_of_week", 1.0), ("month_of_year", 3.0),
                                             ("year", 2022.0), ("second", 1648461600.0)])
def test_local_time_mappers(mk, mapper, expected):
    b = mk(single_feature_config({"name": "x", "type": "local_time", "source": "ranking.localts", "parse": mapper}))
    ev = ranking_event(["p1", "p2"], fields=[{"name": "localts", "value": "2022-03-28T12:00:00+02:00"}])
    eq(b.matrix(ev), [[expected], [expected]])


def test_local_time_errors_and_native_timestamp(mk):
    b = mk(single_feature_config({"name": "x", "type": "local_time", "source": "ranking.localts", "parse": "time_of_day"}))
    eq(b.matrix(ranking_event(["p1"], fields=[{"name": "localts", "value": "now"}])), [[NAN]])
    eq(b.matrix(ranking_event(["p1"])), [[NAN]])
    b2 = mk(single_feature_config({"name": "x", "type": "local_time", "source": "ranking.timestamp", "parse": "year"}))
    eq(b2.matrix(ranking_event(["p1"], timestamp=1648461600000)), [[2022.0]])
    b3 = mk(single_feature_config({"name": "x", "type": "local_time", "source": "ranking.timestamp", "parse": "time_of_day"}))
    eq(b3.matrix(ranking_event(["p1"], timestamp=1661345221008)), [[(12 * 3600 + 47 * 60 + 1) / 3600.0]])  # 2022-08-24T12:47:01Z


@pytest.mark.gpu
def test_local_time_with_a_region_id_through_the_rank_path():
    """"...+01:00[Europe/Paris]" (ZonedDateTime.parse resolves the local time from the REGION's rules at the written instant): the
    host part of the request resolves it from the zoneinfo directory (csrc/tzif.cpp; here the tzdata wheel's), the matrix carries
    it.  The C++ oracle has no tz database - the expectation is computed with Python's zoneinfo (tests/test_local_time_zones_cpu.py
    sweeps zones and instants on the host)."""
    import datetime as dt
    import os
    tzdata = pytest.importorskip("tzdata")
    zoneinfo = pytest.importorskip("zoneinfo")
    from metarank_amd import _native as N
    os.environ["MRK_TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
    N.lib().mrk_debug_tz_reset()
    b = make_backend("hip", single_feature_config({"name": "x", "type": "local_time", "source": "ranking.localts", "parse": "time_of_day"}), "random")
    try:
        for iso, instant, zone in (("2022-03-28T12:00:00+02:00[Europe/Paris]", dt.datetime(2022, 3, 28, 10, tzinfo=dt.timezone.utc), "Europe/Paris"),
                                   ("2022-01-28T12:00:00+02:00[Europe/Paris]", dt.datetime(2022, 1, 28, 10, tzinfo=dt.timezone.utc), "Europe/Paris"),   # written offset is not the zone's: 11:00 there
                                   ("2031-11-02T05:59:59Z[America/New_York]", dt.datetime(2031, 11, 2, 5, 59, 59, tzinfo=dt.timezone.utc), "America/New_York"),
                                   ("2031-11-02T06:00:00Z[America/New_York]", dt.datetime(2031, 11, 2, 6, tzinfo=dt.timezone.utc), "America/New_York")):
            z = instant.astimezone(zoneinfo.ZoneInfo(zone))
            exp = (z.hour * 3600 + z.minute * 60 + z.second) / 3600.0
            eq(b.matrix(ranking_event(["p1", "p2"], fields=[{"name": "localts", "value": iso}])), [[exp], [exp]])
        eq(b.matrix(ranking_event(["p1"], fields=[{"name": "localts", "value": "2022-03-28T12:00:00+02:00[Europe/Atlantis]"}])), [[NAN]])
    finally:
        b.close()
        os.environ.pop("MRK_TZDIR", None)
        N.lib().mrk_debug_tz_reset()


# ---- T/feature/PositionFeatureTest.scala:24-31 ------------------------------------------------------------------------
def test_position_is_constant_online(mk):
    b = mk(single_feature_config({"name": "pos", "type": "position", "position": 5}))
    eq(b.matrix(ranking_event(["p1", "p2", "p3"])), [[5.0], [5.0], [5.0]])


def test_relevancy(mk):
    b = mk(single_feature_config({"name": "rel", "type": "relevancy"}))
    ev = ranking_event([{"id": "p1", "relevancy": 2.5}, {"id": "p2"}, {"id": "p3", "fields": [{"name": "relevancy", "value": 1}]}])
    eq(b.matrix(ev), [[2.5], [NAN], [1.0]])


# ---- T/main/api/RankApiTest.scala:23-29 : constant scores keep the request order (stable sort) -------------------------
def test_rerank_keeps_order_with_noop_scores(mk):
    b = mk(single_feature_config(POP))
    m, scores, order = b.rerank(ranking_event(["p1", "p2", "p3"]))
    assert scores.tolist() == [0.0, 0.0, 0.0] and order.tolist() == [0, 1, 2]


# ---- empty / ragged -----------------------------------------------------------------------------------------------------
def test_duplicate_items_and_single_item(mk):
    b = mk(single_feature_config(POP))
    b.put_double("item=p1/popularity", 7.0)
    eq(b.matrix(ranking_event(["p1", "p1", "p2", "p1"])), [[7.0], [7.0], [NAN], [7.0]])
    eq(b.matrix(ranking_event(["p1"])), [[7.0]])
