"""BASELINE config 1 (plumbing): a 240x180 stream as a DAVIS events.txt, read back,
pumped through the evaluator's window rule (tools/evaluator/src/evaluator.cpp:32-45: compensate when
>= 300000 us since the last compensation or >= 15000 events), ONE patch = the whole frame
(patchCompensateSize = imageSize), 10k-event windows.  The CPU leg uses the oracle (the reference's
CPU path); the GPU leg runs the same file through the HIP path."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def window_rule(ev, freq_time=300000, freq_events=15000):
    """Evaluator::eventCallback's trigger, returning [begin, end) index ranges of the windows."""
    out, begin, last_comp = [], 0, 0
    for i in range(len(ev)):
        n = i + 1 - begin
        if int(ev["t_us"][i]) - last_comp >= freq_time or n >= freq_events:
            out.append((begin, i + 1))
            last_comp = int(ev["t_us"][i])  # lastCompensation = events.back().timestamp (:307)
            begin = i + 1
    return out


@pytest.fixture(scope="module")
def stream_file(tmp_path_factory, synth):
    cfg = dict(synth.CONFIGS[1], index=1)
    evs = []
    for w in range(3):
        e, _ = synth.make_window(cfg, window=w, n_events=10000, vmax=0.3)
        evs.append(e)
    ev = np.concatenate(evs)
    path = tmp_path_factory.mktemp("davis") / "events.txt"
    synth.write_events_txt(str(path), ev)
    return str(path), ev


def test_events_txt_round_trip_and_reader_fixture(ebo, orc, stream_file):
    path, ev = stream_file
    got = ebo.read_events_txt(path)
    rc, ref = orc.parse_events_txt(path)
    assert rc == 0
    assert np.array_equal(got, ref)  # product parser == oracle parser, every field
    assert np.array_equal(got["x"], ev["x"]) and np.array_equal(got["y"], ev["y"])
    assert np.array_equal(got["sign"], ev["sign"])
    assert np.abs(got["t_us"] - ev["t_us"]).max() <= 1  # %.9f seconds -> double -> truncation
    # the reference's own reader fixture (davis240c_reader_test.cpp:19-48)
    fx = ebo.read_events_txt(os.path.join(HERE, "golden", "davis_events_fixture.txt"))
    assert fx["x"].tolist() == [33, 158, 88, 174, 112]
    assert fx["y"].tolist() == [39, 145, 143, 154, 139]
    assert fx["sign"].tolist() == [1, 1, -1, -1, 1]
    assert fx["t_us"].tolist() == [0, 11, 50, 55, 80]


def test_events_txt_in_pieces(ebo, stream_file, tmp_path):
    """ebo_read_events_txt_at: a recording read in pieces of any size (Davis240cReader::getEvents reads 1 000 000 lines
    per call and continues behind them) gives the whole-file parse; the offset ends at the file's size; a piece that stops
    in front of a bad line keeps what it parsed and reports EBO_ERR_RANGE; a last line without a newline is taken."""
    path, _ = stream_file
    whole = ebo.read_events_txt(path)
    for cap in (1, 7, 1000, len(whole), len(whole) + 5):
        off, parts = 0, []
        for _ in range(len(whole) // cap + 3):
            ev, off = ebo.read_events_txt_at(path, off, cap)
            if len(ev) == 0:
                break
            assert len(ev) <= cap
            parts.append(ev)
        assert np.array_equal(np.concatenate(parts), whole), cap
        assert off == os.path.getsize(path)
        if cap >= 1000:
            break_after = len(parts)
            assert break_after == -(-len(whole) // cap)
    p = tmp_path / "tail.txt"
    p.write_text("0.000001 1 2 1\n0.000002 3 4 0")  # no newline at the end
    ev, off = ebo.read_events_txt_at(str(p), 0, 10)
    assert ev["x"].tolist() == [1, 3] and off == os.path.getsize(p)
    q = tmp_path / "bad.txt"
    q.write_text("0.000001 1 2 1\n0.000002 3 4 5\n")
    with pytest.raises(ebo.EboError):
        ebo.read_events_txt_at(str(q), 0, 10)


def test_binary_sidecar_round_trip(ebo, stream_file, tmp_path):
    """The packed sidecar gives back exactly what the text reader parsed."""
    import time
    path, _ = stream_file
    t0 = time.perf_counter()
    parsed = ebo.read_events_txt(path)
    t_txt = time.perf_counter() - t0
    side = str(tmp_path / "events.ebo")
    ebo.write_events_bin(side, parsed)
    assert os.path.getsize(side) == 32 + 16 * len(parsed)
    t0 = time.perf_counter()
    back = ebo.read_events_bin(side)
    t_bin = time.perf_counter() - t0
    assert np.array_equal(back, parsed)
    assert np.array_equal(ebo.read_events_bin(side, cap=7), parsed[:7])  # capacity respected
    # the reference's own fixture through the sidecar
    fx = ebo.read_events_txt(os.path.join(HERE, "golden", "davis_events_fixture.txt"))
    ebo.write_events_bin(side, fx)
    assert np.array_equal(ebo.read_events_bin(side), fx)
    # truncated file, bad magic, unrepresentable event
    raw = open(side, "rb").read()
    open(side, "wb").write(raw[:-5])
    with pytest.raises(ebo.EboError) as ei:
        ebo.read_events_bin(side)
    assert ei.value.code == ebo.ERR_RANGE
    open(side, "wb").write(b"NOTEBOEV" + raw[8:])
    with pytest.raises(ebo.EboError):
        ebo.read_events_bin(side)
    bad = fx.copy()
    bad["x"][0] = 70000
    with pytest.raises(ebo.EboError):
        ebo.write_events_bin(side, bad)
    print("events.txt parse %.1f Mevents/s, sidecar read %.1f Mevents/s"
          % (len(parsed) / t_txt / 1e6, len(parsed) / max(t_bin, 1e-9) / 1e6))


def test_reader_rejects_bad_sign(ebo, tmp_path):
    p = tmp_path / "events.txt"
    p.write_text("0.000001 1 2 1\n0.000002 3 4 7\n")
    with pytest.raises(ebo.EboError) as ei:
        ebo.read_events_txt(str(p))
    assert ei.value.code == ebo.ERR_RANGE


def test_config1_cpu_reference_path(ebo, orc, stream_file):
    """CPU only: reader -> window rule -> oracle compensateEventsContrast + integrateEvents."""
    path, _ = stream_file
    ev = ebo.read_events_txt(path)
    wins = window_rule(ev)
    # the stream starts at t = 1 s: the very first event is already 300 ms past lastCompensation = 0
    assert wins[0] == (0, 1)
    assert [b - a for a, b in wins[1:]] == [15000]  # then the 15000-event cap triggers once
    prm = orc.default_params(patch_w=240, patch_h=180, loss=1)
    assert orc.grid(prm) == (1, 1)
    a, b = wins[1]
    flows, img, s = orc.compensate_events_contrast(ev[a:b], prm, orc.default_solver(max_num_iterations=8))
    assert flows.shape == (1, 2) and np.isfinite(flows).all()
    assert s.final_cost <= s.initial_cost
    assert img.sum() <= b - a
    assert orc.integrate_events(ev[a:b], 240, 180).sum() == b - a
    # a one-event window has no data term (1 <= compensateMinNumEvents): flows stay 0
    f0, _, s0 = orc.compensate_events_contrast(ev[0:1], prm, orc.default_solver())
    assert np.all(f0 == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("loss", [1, 0])
def test_config1_through_the_hip_path(ebo, orc, stream_file, loss):
    path, _ = stream_file
    ev = ebo.read_events_txt(path)
    a, b = window_rule(ev)[1]
    iters = 8
    with ebo.Context(patch_w=240, patch_h=180, loss=loss, max_events=15000) as c:
        assert (c.npx, c.npy) == (1, 1)
        flows, img, s = c.compensate_events_contrast(ev[a:b], ebo.default_solver(max_num_iterations=iters))
        prm = orc.default_params(patch_w=240, patch_h=180, loss=loss)
        fo, io, so = orc.compensate_events_contrast(ev[a:b], prm, orc.default_solver(max_num_iterations=iters))
        assert np.abs(flows - fo).max() <= 1e-5
        assert s.iterations == so.iterations
        assert np.array_equal(img, orc.final_count_image(ev[a:b], prm, flows))
        integ = c.count_image(ebo.COUNT_INTEGRATED)[0]
        assert np.array_equal(integ, orc.integrate_events(ev[a:b], 240, 180))


def test_threaded_reader_entry_point_on_the_reference_fixture(ebo, tmp_path):
    """ebo_read_events_txt_threads on the reference's own fixture (davis240c_reader_test.cpp:19-48's five events) with 1
    and 8 threads asked (a five-line file is parsed by one), and on a file large enough for several threads against
    the default entry point."""
    fx = os.path.join(HERE, "golden", "davis_events_fixture.txt")
    want = ebo.read_events_txt(fx)
    assert len(want) == 5
    for threads in (1, 8):
        got, _, used = ebo.read_events_txt_threads(fx, 16, threads)
        assert used == 1 and np.array_equal(got, want)
    rng = np.random.default_rng(3)
    n = 120_000
    t = 5.0 + np.cumsum(rng.integers(0, 40, n)) * 1e-6
    p = tmp_path / "events.txt"
    with open(p, "w") as f:
        f.write("".join("%.6f %d %d %d\n" % (a, b, c, d) for a, b, c, d in
                        zip(t, rng.integers(0, 240, n), rng.integers(0, 180, n), rng.integers(0, 2, n))))
    whole = ebo.read_events_txt(str(p))
    assert len(whole) == n
    got, off, used = ebo.read_events_txt_threads(str(p), n, 4, offset=0)
    assert used >= 2 and off == os.path.getsize(p) and np.array_equal(got, whole)
    one, _, used1 = ebo.read_events_txt_threads(str(p), n, 1)
    assert used1 == 1 and np.array_equal(one, whole)
    # microseconds: the seconds column through a double, truncated (duration_cast) -- an independent restatement
    assert np.array_equal(whole["t_us"], (np.array([float("%.6f" % a) for a in t]) * 1e6).astype(np.int64))
