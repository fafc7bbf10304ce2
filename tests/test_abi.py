"""CPU tests of the drop-in boundary: libebo_hip.so loads, exports every symbol
include/ebo.h declares, keeps struct layouts in sync with the ctypes mirror, and
fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ebo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ebo_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("ebo_create", "ebo_destroy", "ebo_set_window", "ebo_eval", "ebo_solve",
                 "ebo_count_image", "ebo_compensate_events_contrast", "ebo_patch_integrate",
                 "ebo_patch_integrate_mc", "ebo_last_error", "ebo_shard_range"):
        assert must in syms


def test_library_exports_every_declared_symbol(ebo):
    lib = ebo.lib()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_library_has_gfx950_code_object(ebo):
    blob = open(ebo.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert "gfx950" in ebo.version()


def test_struct_layouts_match_header(ebo, orc):
    assert ebo.EVENT_DTYPE.itemsize == 24 and orc.EVENT_DTYPE == ebo.EVENT_DTYPE
    assert ebo.TRACK_DTYPE.itemsize == 32  # ebo_track_point {int64 id, int64 t_us, double x, y}
    p = ebo.default_params()
    assert (p.image_w, p.image_h, p.patch_w, p.patch_h) == (240, 180, 20, 20)  # feature_detector.h:17,25
    assert (p.tv_weight, p.tv_huber, p.scale, p.min_events) == (1e3, 10.0, 1e-3, 100)  # :26-29
    assert p.loss == ebo.LOSS_EDGE and p.grad == ebo.GRAD_JET
    assert (p.k.max_possible_residual, p.k.sigma_compensate, p.k.kernel_compensate) == (1e3, 1.0, 3)
    assert (p.k.sigma_st, p.k.kernel_st, p.k.kernel_nms) == (1.5, 3, 2)  # contrast_functor.h:282-291
    o = ebo.default_solver()
    assert (o.max_num_iterations, o.use_nonmonotonic) == (50, 1)  # feature_detector.cpp:406-407
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-12,) * 3
    # the oracle's defaults are the same numbers
    q = orc.default_params()
    assert (q.tv_weight, q.tv_huber, q.scale, q.min_events) == (p.tv_weight, p.tv_huber, p.scale, p.min_events)


def test_shard_range_partitions(ebo):
    for n in (0, 1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 4, 8):
            spans = [ebo.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ebo.EboError):
        ebo.shard_range(8, 2, 2)


def test_bad_arguments_are_status_codes(ebo):
    lib = ebo.lib()
    h = C.c_void_p()
    assert lib.ebo_create(None, C.byref(h)) == ebo.ERR_ARG
    p = ebo.default_params(patch_w=0)
    assert lib.ebo_create(C.byref(p), C.byref(h)) == ebo.ERR_ARG
    p = ebo.default_params()
    p.k.kernel_compensate = 5
    assert lib.ebo_create(C.byref(p), C.byref(h)) == ebo.ERR_UNSUPPORTED
    assert b"kernel sizes" in lib.ebo_last_error(None)
    assert lib.ebo_set_window(None, None, 0) == ebo.ERR_ARG
    n = C.c_int(-1)
    assert lib.ebo_device_count(C.byref(n)) == 0 and n.value >= 0


def test_no_gpu_means_no_context(ebo):
    """Without a device the product refuses to run: there is no CPU path."""
    if ebo.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ebo.EboError) as ei:
        ebo.Context()
    assert ei.value.code == ebo.ERR_NO_DEVICE


def test_synthetic_generator_is_deterministic(synth):
    a, fa = synth.make_window(2, n_events=2000)
    b, fb = synth.make_window(2, n_events=2000)
    assert np.array_equal(a, b) and np.array_equal(fa, fb)
    assert np.all(np.diff(a["t_us"]) >= 0)
    assert a["x"].min() >= 0 and a["x"].max() < 240 and a["y"].max() < 180
    assert set(np.unique(a["sign"])) == {-1, 1}
    # first SplitMix64 outputs for seed 0 (published test vector of the generator)
    z = synth.splitmix64(0, 3)
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    npx, npy, rects = synth.grid_rects((240, 180), (30, 22))
    assert (npx, npy) == (8, 8) and tuple(rects[-1]) == (210, 154, 30, 26)


def test_band_plan_rows(ebo):
    """Host only: the plan of every rank from the same row bounds."""
    b = ebo.band_plan(720, [0, 88, 176, 264, 720], 1, 32)
    assert (b.band_row0, b.own_row0, b.own_row1, b.band_row1, b.recv_above, b.recv_below) == (56, 88, 176, 208, 32, 32)
    b = ebo.band_plan(720, [0, 88, 176, 264, 720], 0, 32)
    assert (b.band_row0, b.own_row0, b.own_row1, b.band_row1, b.recv_above, b.recv_below) == (0, 0, 88, 120, 0, 32)
    b = ebo.band_plan(720, [0, 88, 176, 264, 720], 3, 32)
    assert (b.band_row0, b.band_row1, b.recv_above, b.recv_below) == (232, 720, 32, 0)
    b = ebo.band_plan(180, [0, 180], 0, 50)  # one rank: the band is the image
    assert (b.band_row0, b.band_row1, b.recv_above, b.recv_below) == (0, 180, 0, 0)
    with pytest.raises(ebo.EboError):
        ebo.band_plan(180, [0, 100, 170], 0, 5)  # bounds do not end at the image height
