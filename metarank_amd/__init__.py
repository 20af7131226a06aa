"""metarank_amd — MI355X-native /rank hot path for Metarank (feature assembly + LambdaMART scoring).

The product is libmrk_hip.so (metarank_amd/csrc, C ABI in include/mrk.h); this package is the
thin host-side mirror of the reference's interfaces used by tests and benchmarks.
"""
from ._native import MrkError, build, lib, reload_switches  # noqa: F401
from .booster import LIGHTGBM, XGBOOST, Context, HipBooster, default_context  # noqa: F401
from .ranker import Batch, HipRanker, Server  # noqa: F401,E402
from .request import Request, RequestSet  # noqa: F401,E402
