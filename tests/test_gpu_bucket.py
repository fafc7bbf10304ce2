"""GPU tests of the device-side bucketing: the unit table and every
downstream result are identical to the host counting sort's (EBO_BUCKET=host), and to the
oracle's literal O(P*N) per-patch scan (feature_detector.cpp:348-355)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build(ebo, synth, config, ev, offsets, **kw):
    cfg = synth.CONFIGS[config]
    c = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                    patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, max_windows=len(offsets) - 1,
                    max_events=len(ev), **kw)
    return c


@pytest.mark.parametrize("config,n_windows,n_events", [(0, 1, 15000), (2, 5, 20000), (4, 2, 150000)])
def test_device_bucketing_equals_host_bucketing(ebo_ab, orc, synth, monkeypatch, config, n_windows, n_events):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    ev, offsets, gt = synth.make_stream(config, n_windows, n_events=n_events)
    ev["x"][3] = -2  # strays
    ev["y"][11] = 30000 // 2
    results = {}
    for mode in ("host", "device"):
        monkeypatch.setenv("EBO_BUCKET", mode)
        with build(ebo, synth, config, ev, offsets) as c:
            c.set_windows(ev, offsets)
            info = [[c.patch_info(p, w) for p in range(c.P)] for w in range(n_windows)]
            wins = [c.window_info(w) for w in range(n_windows)]
            r, J = c.eval(gt * 0.5)
            img = c.count_image(ebo.COUNT_WARPED, gt * 0.7)
            integ = c.count_image(ebo.COUNT_INTEGRATED)
            results[mode] = (info, wins, r, J, img, integ)
    h, d = results["host"], results["device"]
    assert h[0] == d[0]  # per-patch counts, active flags, reference times
    assert h[1] == d[1]  # per-window reference times and sizes
    # exact fixed-point accumulation: the order inside a unit does not matter, bit for bit
    assert np.array_equal(h[2], d[2]) and np.array_equal(h[3], d[3])
    assert np.array_equal(h[4], d[4]) and np.array_equal(h[5], d[5])
    # and both equal the oracle's literal scan
    cfg = synth.CONFIGS[config]
    prm = orc.default_params(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                             patch_h=cfg["patch"][1], loss=1)
    sub = ev[int(offsets[0]):int(offsets[1])]
    ro, Jo, active, counts = orc.window_eval(sub, prm, gt[0] * 0.5)
    assert [i[0] for i in d[0][0]] == counts.tolist()
    np.testing.assert_allclose(d[2][0], ro, rtol=1e-9)


@pytest.mark.parametrize("quantum_us", [1000, 50000])
def test_quantised_timestamps_keep_results_and_time(ebo_ab, orc, synth, monkeypatch, quantum_us):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Recordings with quantised stamps (a simulator's frame times, a millisecond driver): hundreds to
    thousands of events of a unit share ONE timestamp.  The closed-form canonical order of k_bucket_canon walks
    a run of equal stamps once per record (O(run^2)); runs above 32 records take the bitonic network instead --
    same unit tables, evaluations and images as the host counting sort, bit for bit, and no time cliff (a
    whole window at one stamp used to cost ~10^5 serial LDS reads per thread)."""
    import time
    ev, offsets, gt = synth.make_stream(2, 4, n_events=50000)
    ev = ev.copy()
    ev["t_us"] = (ev["t_us"] // quantum_us) * quantum_us  # 50 ms windows: 50 stamps, or ONE stamp per window
    results, took = {}, {}
    for mode in ("host", "device"):
        monkeypatch.setenv("EBO_BUCKET", mode)
        with build(ebo, synth, 2, ev, offsets) as c:
            c.set_windows(ev, offsets)
            t0 = time.perf_counter()
            c.set_windows(ev, offsets)
            took[mode] = time.perf_counter() - t0
            info = [[c.patch_info(p, w) for p in range(c.P)] for w in range(4)]
            r, J = c.eval(gt * 0.5)
            img = c.count_image(ebo.COUNT_WARPED, gt * 0.7)
            results[mode] = (info, r, J, img)
    h, d = results["host"], results["device"]
    assert h[0] == d[0]
    assert np.array_equal(h[1], d[1]) and np.array_equal(h[2], d[2]) and np.array_equal(h[3], d[3])
    assert took["device"] < 0.05, took  # 200 k events: ~1 ms; the quadratic walk took tens of milliseconds per unit


def test_device_resident_raw_events(ebo, orc, synth):
    """ebo_set_windows_device: the raw 24-byte records never touch the host path."""
    import ctypes
    ev, offsets, gt = synth.make_stream(2, 4, n_events=25000)
    # device buffer through the HIP runtime the library itself uses (no second runtime)
    hip = ctypes.CDLL("libamdhip64.so")
    d_raw = ctypes.c_void_p()
    host = np.ascontiguousarray(ev)
    assert hip.hipMalloc(ctypes.byref(d_raw), ctypes.c_size_t(host.nbytes)) == 0
    assert hip.hipMemcpy(d_raw, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(host.nbytes), 1) == 0

    class _Raw:
        def data_ptr(self):
            return d_raw.value
    raw = _Raw()
    with build(ebo, synth, 2, ev, offsets) as c:
        c.set_windows_device(raw.data_ptr(), offsets)
        r, J = c.eval(gt * 0.5)
        prm = orc.default_params(patch_w=30, patch_h=22, loss=1)
        for w in range(4):
            sub = ev[int(offsets[w]):int(offsets[w + 1])]
            ro, Jo, _, _ = orc.window_eval(sub, prm, gt[w] * 0.5)
            np.testing.assert_allclose(r[w], ro, rtol=1e-9)
            np.testing.assert_allclose(J[w], Jo, rtol=1e-9, atol=1e-10)
            assert c.window_info(w) == (orc.mid_timestamp(sub["t_us"][0], sub["t_us"][-1]), len(sub))
        # a sub-range of the device buffer (offsets need not start at 0)
        c.set_windows_device(raw.data_ptr(), offsets[1:3])
        r1, _ = c.eval(gt[1:2] * 0.5)
        np.testing.assert_array_equal(r1[0], r[1])


def test_device_bucketing_error_flags(ebo, synth):
    ev, _ = synth.make_window(0, n_events=3000)
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        bad = ev.copy()
        bad["y"][7] = -20000
        with pytest.raises(ebo.EboError) as ei:
            c.set_window(bad)
        assert ei.value.code == ebo.ERR_RANGE and "coordinate" in str(ei.value)
        far = ev.copy()
        far["t_us"][-1] += 1 << 33
        with pytest.raises(ebo.EboError) as ei:
            c.set_window(far)
        assert ei.value.code == ebo.ERR_RANGE
        late = ev.copy()
        late["t_us"] += 1 << 32  # mid time beyond int32: undefined in the reference
        with pytest.raises(ebo.EboError) as ei:
            c.set_window(late)
        assert ei.value.code == ebo.ERR_RANGE and "int32" in str(ei.value)
        c.set_window(ev)
        assert c.eval(np.zeros((c.P, 2)))[0].shape == (1, c.P)


@pytest.mark.parametrize("config,n_windows,n_events", [(0, 1, 15000), (2, 9, 20000), (3, 20, 120000), (4, 3, 150000)])
def test_compact_records_give_the_same_windows(ebo, synth, config, n_windows, n_events):
    """ebo_set_windows8 (8-byte records, per-window base times, upload pipelined with the bucketing in
    groups of windows) against ebo_set_windows (24-byte EventSamples) on the same events: the same
    unit tables, reference times, objective values, Jacobians and count images, bit for bit; from
    host memory and from device memory; with strays; with a base time that is not the first event's."""
    import torch
    ev, offsets, gt = synth.make_stream(config, n_windows, n_events=n_events)
    ev["x"][3] = -2  # strays
    ev["y"][11] = 15000
    t_base = np.array([int(ev["t_us"][int(offsets[w])]) - 7 * w for w in range(n_windows)], dtype=np.int64)
    ev8 = np.concatenate([ebo.pack_events8(ev[int(offsets[w]):int(offsets[w + 1])], t_base[w]) for w in range(n_windows)])
    assert ev8.dtype.itemsize == 8 and len(ev8) == len(ev)

    def snapshot(c):
        info = [[c.patch_info(p, w) for p in range(c.P)] for w in range(n_windows)]
        wins = [c.window_info(w) for w in range(n_windows)]
        r, J = c.eval(gt * 0.5)
        return info, wins, r, J, c.count_image(ebo.COUNT_WARPED, gt * 0.7), c.count_image(ebo.COUNT_INTEGRATED)

    with build(ebo, synth, config, ev, offsets) as c:
        c.set_windows(ev, offsets)
        ref = snapshot(c)
        c.set_windows8(ev8, t_base, offsets)
        a = snapshot(c)
        d8 = torch.from_numpy(ev8.view(np.uint8).reshape(-1, 8)).to("cuda")
        c.set_windows8(d8.data_ptr(), t_base, offsets, device=True)
        b = snapshot(c)
        pinned = torch.from_numpy(ev8.view(np.uint8).reshape(-1, 8)).pin_memory()
        c.set_windows8(pinned.data_ptr(), t_base, offsets)
        p = snapshot(c)
    for got in (a, b, p):
        assert got[0] == ref[0] and got[1] == ref[1]
        for k in range(2, 6):
            assert np.array_equal(got[k], ref[k])


def test_compact_record_range_errors(ebo):
    ev = ebo.make_events([1, 2], [3, 4], [10, 5_000_000_000])
    with pytest.raises(ebo.EboError) as ei:
        ebo.pack_events8(ev, 0)  # 5e9 us does not fit int32
    assert ei.value.code == ebo.ERR_RANGE
    ev = ebo.make_events([1, 20000], [3, 4], [10, 20])
    with pytest.raises(ebo.EboError):
        ebo.pack_events8(ev, 0)  # x beyond 15 bits
    ok = ebo.pack_events8(ebo.make_events([-3, 5], [7, -1], [100, 90], sign=[1, -1]), 95)
    assert ok["t_rel_us"].tolist() == [5, -5]
    assert (ok["xy"] & 0x7FFF).tolist() == [(-3) & 0x7FFF, 5] and ((ok["xy"] >> 15) & 1).tolist() == [1, 0]
