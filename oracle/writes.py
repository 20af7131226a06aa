"""CPU oracle of the WRITE side that feeds the /rank path (test infrastructure only - never imported by the
product).  Restates, in plain Python integers, the raw-state primitives of the reference:

  * MemPeriodicCounter.put / computeValue      fstore/memory/MemPeriodicCounter.scala:16-37
    Timestamp.toStartOfPeriod                  model/Timestamp.scala:18-21  (floor of a DOUBLE division)
    PeriodicCounterFeature.fromMap             model/Feature.scala:142-161
  * MemBoundedList.put                         fstore/memory/MemBoundedList.scala:18-37
  * MemCounter.put                             fstore/memory/MemCounter.scala

Pinned on the reference's own known-answer suites (tests/test_write_path.py): fstore/PeriodicCounterSuite.scala:22-144
(4 cases), fstore/BoundedListSuite.scala:24-68 (4 cases), feature/WindowInteractionCountFeatureTest.scala:46-57,
feature/RateFeatureTest.scala:61-74.
"""
from __future__ import annotations

import math

DUR = {"s": 1000, "m": 60_000, "h": 3_600_000, "d": 86_400_000}


def duration_ms(s: str) -> int:
    """util/DurationJson.scala:9-13: ([0-9]+)([smhd])"""
    return int(s[:-1]) * DUR[s[-1]]


def start_of_period(ts: int, period_ms: int) -> int:
    """Timestamp.toStartOfPeriod: math.floor(ts.toDouble / period.toMillis).toLong * period.toMillis"""
    return int(math.floor(float(ts) / float(period_ms))) * period_ms


class PeriodicCounter:
    """one key of a PeriodicCounterFeature: Map[bucket start -> count]"""

    def __init__(self, period_ms: int, ranges):
        self.period = period_ms
        self.ranges = list(ranges)  # [(startOffset, endOffset)]
        self.buckets: dict[int, int] = {}

    def put(self, ts: int, inc: int):
        b = start_of_period(ts, self.period)
        self.buckets[b] = self.buckets.get(b, 0) + inc

    def values(self):
        """fromMap -> [(start, end, periods, sum)]; None when nothing was ever put (computeValue -> None)"""
        if not self.buckets:
            return None
        last = max(self.buckets)  # map.ts.lastOption of the sorted timestamps: the anchor is the LATEST bucket present
        out = []
        for so, eo in self.ranges:
            start = last - self.period * so
            end = last - self.period * eo + self.period
            total = sum(c for b, c in self.buckets.items() if start <= b <= end)
            out.append((start, end, so - eo + 1, total))
        return out

    def sums(self):
        v = self.values()
        return None if v is None else [x[3] for x in v]


class BoundedList:
    """one key of a BoundedListFeature: List[TimeValue], newest first"""

    def __init__(self, count: int, duration_ms_: int):
        self.count = count
        self.duration = duration_ms_
        self.items: list[tuple[int, str]] | None = None

    def put(self, value: str, ts: int):
        if self.items is None:
            self.items = [(ts, value)]  # the first element is stored without any filtering
            return
        result = [(ts, value)] + self.items
        cutoff = ts - self.duration
        self.items = [e for e in result if e[0] >= cutoff][: self.count]

    def values(self):
        return None if self.items is None else [v for _, v in self.items]


class WriteState:
    """Raw state of every key + the configs of the states a Metarank `features:` section declares."""

    def __init__(self, config: dict):
        self.periodic_cfg: dict[str, tuple[int, list]] = {}   # state name -> (period, ranges)
        self.list_cfg: dict[str, tuple[int, int]] = {}        # state name -> (count, duration)
        for f in config["features"]:
            t, name = f["type"], f["name"]
            if t == "window_count":
                self.periodic_cfg[name] = (duration_ms(f["bucket"]), [(p, 0) for p in f["periods"]])
            elif t == "rate":
                cfg = (duration_ms(f["bucket"]), [(p, 0) for p in f["periods"]])
                for s in (f"{name}_{f['top']}", f"{name}_{f['bottom']}", f"{name}_{f['top']}_norm", f"{name}_{f['bottom']}_norm"):
                    self.periodic_cfg[s] = cfg
            elif t == "interacted_with":
                self.list_cfg[f"{name}_interactions"] = (int(f.get("count", 100)), duration_ms(f.get("duration", "24h")))
        self.periodic: dict[str, PeriodicCounter] = {}
        self.lists: dict[str, BoundedList] = {}
        self.counters: dict[str, int] = {}

    @staticmethod
    def _state(key: str) -> str:
        return key.split("/", 1)[1]

    def increment_periodic(self, key: str, ts: int, inc: int = 1):
        c = self.periodic.get(key)
        if c is None:
            period, ranges = self.periodic_cfg[self._state(key)]
            c = self.periodic[key] = PeriodicCounter(period, ranges)
        c.put(ts, inc)
        return c.sums()

    def increment(self, key: str, inc: int = 1):
        self.counters[key] = self.counters.get(key, 0) + inc
        return self.counters[key]

    def append(self, key: str, value: str, ts: int):
        l = self.lists.get(key)
        if l is None:
            count, dur = self.list_cfg[self._state(key)]
            l = self.lists[key] = BoundedList(count, dur)
        l.put(value, ts)
        return l.values()
