"""Launch shapes chosen from a batch's shape (metarank_amd/csrc/launch_shape.hpp): host arithmetic, no device.

Each expectation is the choice an A/B measurement on the MI355X ended in (profiles/r02_r_slices.txt,
profiles/r02_ab_runs.json, gpurun_out/r02_l); the test keeps a later edit of the rules from silently moving the
BASELINE configurations onto a shape that was measured slower."""
import ctypes as C

import pytest

from metarank_amd import _native


@pytest.fixture(scope="module")
def lib():
    L = _native.lib()
    L.mrk_debug_fused_shape.restype = C.c_int
    L.mrk_debug_fused_shape.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.mrk_debug_scorer_split.restype = C.c_int
    L.mrk_debug_scorer_split.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    return L


def fused(lib, n_req, max_items):
    out = (C.c_int * 3)()
    assert lib.mrk_debug_fused_shape(n_req, max_items, out) == 0
    return tuple(out)  # (lanes per workgroup, op split, slices per request)


@pytest.mark.parametrize("n_req,max_items,want,why", [
    (3840, 100, (128, 1, 1), "c2: two wavefronts per request (64 lanes = two serial rounds: 0.421 vs 0.288 ms)"),
    (384, 1000, (256, 1, 2), "c3: two workgroups per request fill one residency (0.62 -> 0.43 ms; 3 / 4 slices: 0.47 / 0.48)"),
    (96, 1000, (256, 1, 4), "96 large requests: one workgroup per round of the lanes (0.52 -> 0.21 ms)"),
    (1, 100, (512, 4, 1), "a single /rank request: four copies of the item lanes share the ops (p50 0.19 -> 0.158 ms)"),
    (16, 100, (512, 4, 1), "... up to 16 requests"),
    (64, 100, (512, 4, 1), "... and up to 64 when they are small (mrk_rank's combined batches: 188 k -> 228 k requests/s at 64 callers, r06_x)"),
    (65, 100, (128, 1, 1), "... and not beyond"),
    (17, 1000, (256, 1, 4), "17 large requests keep their slices"),
    (1, 1000, (512, 2, 1), "a single 1 000-item request: 256 item lanes leave room for two copies; no slices on top of a split"),
    (1, 1, (256, 4, 1), "one candidate: one wavefront of item lanes, four copies"),
    (3840, 300, (256, 1, 1), "a full batch of 300-item requests: 2 rounds, but 3 840 x 4 wavefronts already exceed a residency"),
    (4096, 64, (64, 1, 1), "64-item requests: one wavefront each"),
])
def test_fused_kernel_shape(lib, n_req, max_items, want, why):
    assert fused(lib, n_req, max_items) == want, why


def test_slices_never_exceed_the_rounds_or_one_residency(lib):
    for n_req in (17, 32, 100, 384, 1000, 5000):
        for items in (65, 128, 257, 512, 700, 1000, 1024):
            lanes, split, slices = fused(lib, n_req, items)
            rounds = -(-items // (lanes // split))
            assert 1 <= slices <= rounds
            assert slices == 1 or n_req * (lanes // 64) * slices <= 4096


@pytest.mark.parametrize("rows,views,f64,want,why", [
    (384_000, 41, 1, 4, "c2 (V = 41): 1 -> 0.283 ms, 2 -> 0.317, 4 -> 0.218, 8 -> 0.220, 16 -> 0.230"),
    (384_000, 100, 1, 8, "c3-like (V ~ 100): 1 -> 0.547 ms, 4 -> 0.280, 8 -> 0.248"),
    (100, 41, 1, 16, "a single request: one tile, every wavefront the kernel has"),
    (100_000, 41, 1, 8, "c4: 782 tiles (r04_h: 4 -> 0.098 ms, 8 -> 0.083, 16 -> 0.083)"),
    (20_000, 41, 1, 16, "157 tiles (r04_h: 4 -> 0.066 ms, 8 -> 0.045, 16 -> 0.032)"),
    (400_000, 41, 1, 4, "3 125 tiles (r04_h: 4 -> 0.231 ms, 8 -> 0.233, 16 -> 0.242)"),
    (4_000_000, 41, 1, 4, "c4x: 31 250 tiles"),
    (384_000, 0, 1, 1, "a forest of single-leaf trees has no views (and no division by zero: r02_m)"),
])
def test_scorer_wavefronts_per_tile(lib, rows, views, f64, want, why):
    assert lib.mrk_debug_scorer_split(rows, views, f64, 256) == want, why
