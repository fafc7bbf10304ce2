"""GPU tests of the count-image implementations (0: global int atomics, 1: whole-window LDS
bands, 2: patch-row LDS bands with an overflow list for events that leave their band, 3: events
sorted by destination band, 4: unit waves with a displacement bound per unit, 5: unit waves over
2-D tiles, 6: a rolling band per window that reads every event once, the default for warped images in
large launches): all
bit-exact against the oracle, on single windows and on batches, for all three modes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _prm(orc, c):
    p = c.params
    return orc.default_params(image_w=p.image_w, image_h=p.image_h, patch_w=p.patch_w,
                              patch_h=p.patch_h, scale=p.scale, min_events=p.min_events, loss=1)


@pytest.mark.parametrize("impl", ["0", "1", "2", "3", "4", "5", "auto"])
@pytest.mark.parametrize("config,n_windows,n_events", [(0, 1, 15000), (2, 70, 9000), (3, 3, 70000), (4, 2, 60000)])
def test_count_image_implementations(ebo_ab, orc, synth, monkeypatch, impl, config, n_windows, n_events):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    if impl == "auto":
        monkeypatch.delenv("EBO_COUNT_IMPL", raising=False)
    else:
        monkeypatch.setenv("EBO_COUNT_IMPL", impl)
    cfg = synth.CONFIGS[config]
    ev, offsets, gt = synth.make_stream(config, n_windows, n_events=n_events)
    # a few events outside the sensor in the first window (stray bucket)
    ev["x"][5] = -7
    ev["y"][9] = cfg["image"][1] + 3
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                     patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, max_windows=n_windows,
                     max_events=len(ev)) as c:
        c.set_windows(ev, offsets)
        prm = _prm(orc, c)
        w, h = cfg["image"]
        rng = np.random.RandomState(17)
        flows = rng.uniform(-2, 2, (n_windows, c.P, 2))
        field = rng.uniform(-2, 2, (n_windows, h, w, 2)).astype(np.float32)
        integ = c.count_image(ebo.COUNT_INTEGRATED)
        warped = c.count_image(ebo.COUNT_WARPED, flows)
        byfield = c.count_image(ebo.COUNT_FIELD, field)
        again = c.count_image(ebo.COUNT_INTEGRATED)
        for k in sorted(set([0, n_windows // 2, n_windows - 1])):
            sub = ev[int(offsets[k]):int(offsets[k + 1])]
            assert np.array_equal(integ[k], orc.integrate_events(sub, w, h))
            assert np.array_equal(warped[k], orc.final_count_image(sub, prm, flows[k]))
            assert np.array_equal(byfield[k], orc.compensate_events_field(sub, w, h, field[k]))
            assert np.array_equal(again[k], integ[k])


@pytest.mark.parametrize("impl", ["1", "2", "5"])
def test_count_image_more_than_65535_events_per_window(ebo_ab, orc, synth, monkeypatch, impl):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """16-bit packed counters are only used below 65536 events per window."""
    monkeypatch.setenv("EBO_COUNT_IMPL", impl)
    cfg = synth.CONFIGS[2]
    ev, _ = synth.make_window(2, n_events=90000)
    ev["x"][:70000] = 11  # 70000 events on ONE pixel: would overflow a 16-bit counter
    ev["y"][:70000] = 13
    with ebo.Context(image_w=240, image_h=180, patch_w=30, patch_h=22, loss=ebo.LOSS_VARIANCE,
                     max_events=len(ev)) as c:
        c.set_window(ev)
        img = c.count_image(ebo.COUNT_INTEGRATED)[0]
        assert img[13, 11] >= 70000
        assert np.array_equal(img, orc.integrate_events(ev, 240, 180))


@pytest.mark.parametrize("lds_kb", ["12", "24", "150"])
def test_patch_row_bands_any_band_size_and_large_flows(ebo_ab, orc, synth, monkeypatch, lds_kb):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """impl 2 with bands of one patch row up to the whole image, flows large enough that most
    events leave their band (overflow list) or the image."""
    monkeypatch.setenv("EBO_COUNT_IMPL", "2")
    monkeypatch.setenv("EBO_COUNT_LDS_KB", lds_kb)
    cfg = synth.CONFIGS[2]
    ev, offsets, gt = synth.make_stream(2, 5, n_events=20000)
    ev["x"][3] = -2  # stray events, warped back inside by the flow of their clamped patch
    ev["y"][3] = 100
    ev["y"][7] = 185
    with ebo.Context(image_w=240, image_h=180, patch_w=30, patch_h=22, loss=ebo.LOSS_VARIANCE,
                     max_windows=5, max_events=len(ev)) as c:
        c.set_windows(ev, offsets)
        prm = _prm(orc, c)
        rng = np.random.RandomState(3)
        flows = rng.uniform(-8, 8, (5, c.P, 2))
        warped = c.count_image(ebo.COUNT_WARPED, flows)
        integ = c.count_image(ebo.COUNT_INTEGRATED)
        for k in range(5):
            sub = ev[int(offsets[k]):int(offsets[k + 1])]
            assert np.array_equal(warped[k], orc.final_count_image(sub, prm, flows[k]))
            assert np.array_equal(integ[k], orc.integrate_events(sub, 240, 180))


@pytest.mark.parametrize("impl", ["1", "3", "4"])
@pytest.mark.parametrize("image,patch", [((16383, 24), (2, 24)), ((16383, 24), (3, 5)), ((16000, 20), (127, 1)),
                                         ((9000, 30), (8999, 7)), ((40, 16383), (1, 16383)), ((64, 48), (1, 1))])
def test_patch_of_an_event_from_its_coordinates(ebo_ab, orc, monkeypatch, impl, image, patch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """The warped-count kernels find an event's patch as min(x / patch_w, npx - 1) with one
    multiply-high by a precomputed reciprocal: exact over the whole 15-bit coordinate range, for
    divisors from 1 to the sensor size, ragged last patches, and events outside the sensor (which
    take the flow of the clamped patch)."""
    monkeypatch.setenv("EBO_COUNT_IMPL", impl)
    w, h = image
    rng = np.random.RandomState(w + patch[0])
    n = 6000
    x = rng.randint(0, w, n).astype(np.int32)
    y = rng.randint(0, h, n).astype(np.int32)
    # both sides of every patch boundary near the ends, the last pixel, and a few strays
    x[:8] = [0, patch[0] - 1, min(patch[0], w - 1), w - 1, w - 2, max(w - patch[0], 0), -3, min(w + 2, 16383)]
    y[:8] = [0, patch[1] - 1, min(patch[1], h - 1), h - 1, h - 2, max(h - patch[1], 0), 2, min(h + 1, 16383)]
    t = np.sort(rng.randint(0, 40000, n)) + 10_000
    ev = ebo.make_events(x, y, t, np.ones(n, dtype=np.int32))
    with ebo.Context(image_w=w, image_h=h, patch_w=patch[0], patch_h=patch[1], loss=ebo.LOSS_VARIANCE,
                     tv_weight=0.0, max_events=n) as c:
        c.set_window(ev)
        prm = _prm(orc, c)
        flows = rng.uniform(-1.5, 1.5, (c.P, 2))
        warped = c.count_image(ebo.COUNT_WARPED, flows)[0]
        assert np.array_equal(warped, orc.final_count_image(ev, prm, flows))


@pytest.mark.parametrize("impl", ["5"])
@pytest.mark.parametrize("w,h,pw,ph,n_events,lds_kb", [(347, 261, 21, 16, 30000, 16), (347, 261, 21, 16, 70000, 24),
                                                      (64, 48, 7, 5, 4000, 1), (1280, 720, 40, 22, 200000, 0),
                                                      (346, 260, 21, 16, 50000, 0), (240, 180, 30, 22, 20000, 40)])
def test_tiled_count_image_odd_sizes_and_wide_counters(ebo_ab, orc, synth, monkeypatch, impl, w, h, pw, ph, n_events, lds_kb):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """k_count_tiles (impl 5) and k_count_sweep (impl 6) on sizes that exercise their edges: odd image
    widths (no 16-byte row stores, packed counters shared between rows), tiles / strips that do not
    divide the image, 32-bit counters (>= 65536 events per window), tiny tiles and bands (every unit
    reaches several; the rolling band falls back to band-by-band counting), flows large enough to
    leave the image or NaN, stray events."""
    monkeypatch.setenv("EBO_COUNT_IMPL", impl)
    if lds_kb:
        monkeypatch.setenv("EBO_COUNT_LDS_KB", str(lds_kb))
    cfg = dict(name="t", image=(w, h), patch=(pw, ph), events=n_events, index=11)
    n_windows = 3
    ev, offsets, gt = synth.make_stream(cfg, n_windows)
    ev["x"][3] = -5
    ev["y"][7] = h + 2
    ev["x"][11] = w + 40
    with ebo.Context(image_w=w, image_h=h, patch_w=pw, patch_h=ph, loss=ebo.LOSS_VARIANCE, max_windows=n_windows,
                     max_events=len(ev)) as c:
        c.set_windows(ev, offsets)
        prm = _prm(orc, c)
        rng = np.random.RandomState(5)
        flows = rng.uniform(-3, 3, (n_windows, c.P, 2))
        flows[0, 0] = (40.0, -35.0)      # leaves the image
        flows[1, 1] = (np.nan, 1.0)      # undefined in the reference: the event is skipped on both sides
        warped = c.count_image(ebo.COUNT_WARPED, flows)
        monkeypatch.setenv("EBO_COUNT_IMPL", "0")
        plain = c.count_image(ebo.COUNT_WARPED, flows)
        assert np.array_equal(warped, plain)
        for k in range(n_windows):
            sub = ev[int(offsets[k]):int(offsets[k + 1])]
            assert np.array_equal(warped[k], orc.final_count_image(sub, prm, flows[k]))
