"""A SINGLE window sharded over ranks by rows of the patch grid (C4 = 128 patches per
GPU): rank r takes the patches of its rows and the events inside them (ebo_shard_range +
ebo_set_patches); evaluations and per-patch solves of the shards, concatenated in rank order (what
the all-gather delivers), equal the whole window's (values bit for bit: a patch's result depends on its own events only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,world", [(0, 2), (2, 3), (3, 8)])
def test_patch_row_shards_equal_the_whole_window(ebo, synth, config, world):
    cfg = synth.CONFIGS[config]
    ev, gt = synth.make_window(config, n_events=min(cfg["events"], 60000))
    kw = dict(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
              loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev))
    flows = gt * 0.5
    with ebo.Context(**kw) as c:
        c.set_window(ev)
        npx, npy, P = c.npx, c.npy, c.P
        r, J = c.eval(flows)
        solved, _ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=8)
        rects = [c.patch_rect(p % npx, p // npx) for p in range(P)]
        active = [c.patch_info(p)[1] for p in range(P)]
    parts_r, parts_J, parts_s = [], [], []
    for rank in range(world):
        b, e = ebo.shard_range(npy, rank, world)  # rows of the patch grid
        mine = list(range(b * npx, e * npx))
        evs, offs = [], [0]
        for p in mine:
            x0, y0, pw, ph = rects[p]
            sel = (ev["x"] >= x0) & (ev["x"] < x0 + pw) & (ev["y"] >= y0) & (ev["y"] < y0 + ph)
            evs.append(ev[sel])
            offs.append(offs[-1] + int(sel.sum()))
        if not mine:
            continue
        with ebo.Context(**kw) as c:
            c.set_patches(np.concatenate(evs), offs, [rects[p] for p in mine])
            rr, JJ = c.eval(flows[mine])
            ss, _ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=8)
        parts_r.append(rr[0])
        parts_J.append(JJ[0])
        parts_s.append(ss[0])
    rr, JJ, ss = np.concatenate(parts_r), np.concatenate(parts_J), np.concatenate(parts_s)
    act = np.array(active)
    # the reference time of a shard's patch is its own events' (contrast_functor.h:18-20), the image
    # is accumulated exactly: the value is identical; the Jacobian's f64 partial sums may be tiled
    # differently (the row tiling follows the largest rect a context holds): last bits only
    assert np.array_equal(rr[act], r[0][act])
    np.testing.assert_allclose(JJ[act], J[0][act], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ss[act], solved[0][act], rtol=0, atol=1e-9)


@pytest.mark.parametrize("config,world,n_windows,stray", [(0, 2, 2, 0), (2, 3, 1, 40), (3, 8, 2, 25), (4, 8, 1, 0)])
def test_partial_count_images_of_the_shards_sum_to_the_whole_image(ebo, synth, config, world, n_windows, stray):
    """Config 4 end to end (SURVEY 8(e) "final full-frame count image", feature_detector.cpp:433-463): every
    rank counts ITS events (ebo_set_patches + ebo_count_image_shard) warped by the gathered flows of ALL
    patches at the WINDOW's reference time; the partial images are integer-valued, their sum equals the
    one-process image bit for bit -- with events outside the sensor (they take the clamped patch's flow
    and live on that patch's rank), large flows that carry events across shard borders, several windows."""
    cfg = synth.CONFIGS[config]
    iw, ih = cfg["image"]
    pw, ph = cfg["patch"]
    rng = np.random.default_rng(11 + config)
    evs, offs = [], [0]
    for w in range(n_windows):
        ev, _ = synth.make_window(config, window=w, n_events=min(cfg["events"], 120000))
        if stray:
            k = rng.choice(len(ev), stray, replace=False)
            ev = ev.copy()
            ev["x"][k[: stray // 2]] = rng.integers(-30, 0, stray // 2)
            ev["y"][k[stray // 2:]] = ih + rng.integers(0, 30, stray - stray // 2)
        evs.append(ev)
        offs.append(offs[-1] + len(ev))
    allev = np.concatenate(evs)
    kw = dict(image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=ebo.LOSS_VARIANCE, tv_weight=0.0,
              max_events=len(allev), max_windows=n_windows)
    with ebo.Context(**kw) as c:
        c.set_windows(allev, offs)
        npx, npy, P = c.npx, c.npy, c.P
        flows = rng.uniform(-6.0, 6.0, (n_windows, P, 2))  # up to +-150 px over a 50 ms window: far across shard borders
        whole = c.count_image(ebo.COUNT_WARPED, flows)
        rects = np.array([c.patch_rect(p % npx, p // npx) for p in range(P)])
        t_ref = [c.window_info(w)[0] for w in range(n_windows)]
    for w in range(n_windows):
        assert t_ref[w] == ebo.window_ref_time(evs[w]["t_us"][0], evs[w]["t_us"][-1])
    total = np.zeros_like(whole)
    for rank in range(world):
        b, e = ebo.shard_range(npy, rank, world)
        if b == e:
            continue
        my = np.arange(b * npx, e * npx)
        sev, soffs = [], [0]
        for w in range(n_windows):
            ev = evs[w]
            # the patch of the final loop (:436-441): index clamped into the grid, also for stray events
            gx = np.clip(np.trunc(ev["x"] / pw).astype(np.int64), 0, npx - 1)
            gy = np.clip(np.trunc(ev["y"] / ph).astype(np.int64), 0, npy - 1)
            pid = gy * npx + gx
            for p in my:
                sel = ev[pid == p]
                sev.append(sel)
                soffs.append(soffs[-1] + len(sel))
        with ebo.Context(**kw) as c:
            c.set_patches(np.concatenate(sev), soffs, np.tile(rects[my], (n_windows, 1)))
            part = c.count_image_shard(n_windows, t_ref, flows)
        assert np.array_equal(part, np.round(part)) and part.min() >= 0
        total += part
    assert np.array_equal(total, whole)
    assert whole.sum() > 0.5 * len(allev)


def test_count_image_shard_argument_errors(ebo, synth):
    ev, _ = synth.make_window(0, n_events=3000)
    with ebo.Context(max_events=len(ev), max_windows=2) as c:
        c.set_window(ev)
        with pytest.raises(ebo.EboError) as err:  # a window context is not a shard
            c.count_image_shard(1, [0], np.zeros((1, c.P, 2)))
        assert err.value.code == ebo.ERR_STATE
        c.set_patches(ev, [0, 1000, 3000], [(0, 0, 20, 20), (20, 0, 20, 20)])
        with pytest.raises(ebo.EboError) as err:  # 2 units are not 3 equal groups
            c.count_image_shard(3, [0, 0, 0], np.zeros((3, c.P, 2)))
        assert err.value.code == ebo.ERR_ARG
        with pytest.raises(ebo.EboError) as err:  # reference time 2^31 us away from the events
            c.count_image_shard(1, [int(ev["t_us"][0]) + (1 << 32)], np.zeros((1, c.P, 2)))
        assert err.value.code == ebo.ERR_RANGE


def _band_setup(ebo, synth, config, world, n_windows, stray, flow_amp, seed):
    cfg = synth.CONFIGS[config]
    iw, ih = cfg["image"]
    pw, ph = cfg["patch"]
    rng = np.random.default_rng(seed)
    evs, offs = [], [0]
    for w in range(n_windows):
        ev, _ = synth.make_window(config, window=w, n_events=min(cfg["events"], 120000))
        if stray:
            k = rng.choice(len(ev), stray, replace=False)
            ev = ev.copy()
            ev["x"][k[: stray // 2]] = rng.integers(-30, 0, stray // 2)
            ev["y"][k[stray // 2:]] = ih + rng.integers(0, 30, stray - stray // 2)
        evs.append(ev)
        offs.append(offs[-1] + len(ev))
    allev = np.concatenate(evs)
    kw = dict(image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=ebo.LOSS_VARIANCE, tv_weight=0.0,
              max_events=len(allev), max_windows=n_windows)
    with ebo.Context(**kw) as c:
        c.set_windows(allev, offs)
        npx, npy, P = c.npx, c.npy, c.P
        flows = rng.uniform(-flow_amp, flow_amp, (n_windows, P, 2))
        whole = c.count_image(ebo.COUNT_WARPED, flows)
        rects = np.array([c.patch_rect(p % npx, p // npx) for p in range(P)])
        t_ref = [c.window_info(w)[0] for w in range(n_windows)]
    return dict(cfg=cfg, evs=evs, kw=kw, npx=npx, npy=npy, P=P, flows=flows, whole=whole, rects=rects, t_ref=t_ref,
                iw=iw, ih=ih, pw=pw, ph=ph)


def _shard_events(S, b, e, n_windows):
    npx, npy, pw, ph = S["npx"], S["npy"], S["pw"], S["ph"]
    my = np.arange(b * npx, e * npx)
    sev, soffs = [], [0]
    for w in range(n_windows):
        ev = S["evs"][w]
        gx = np.clip(np.trunc(ev["x"] / pw).astype(np.int64), 0, npx - 1)
        gy = np.clip(np.trunc(ev["y"] / ph).astype(np.int64), 0, npy - 1)
        pid = gy * npx + gx
        for p in my:
            sel = ev[pid == p]
            sev.append(sel)
            soffs.append(soffs[-1] + len(sel))
    return my, np.concatenate(sev), soffs


@pytest.mark.parametrize("config,world,n_windows,stray,flow_amp,halo,expect_escape", [
    (0, 2, 2, 0, 0.6, 20, False),     # reference grid, two ranks, reach <= 0.6 * 26 + 1 < 20 rows
    (2, 3, 1, 40, 0.5, 16, False),    # C2 with events outside the sensor (they live on the border patches' ranks)
    (3, 8, 2, 25, 0.9, 28, False),    # C3 over 8 ranks: 2 grid rows = 32 image rows per rank, halo 28
    (4, 8, 1, 0, 1.0, 32, False),     # C4 as the bench shards it: 90 rows per rank, halo 32
    (3, 8, 1, 0, 6.0, 28, True),      # +-150 px: far beyond any halo -> every rank reports it, the dense path answers
    (4, 3, 1, 0, 1.0, 400, None),     # a halo larger than a rank's rows: no plan (ERR_UNSUPPORTED on every rank)
])
def test_band_limited_images_of_the_shards_assemble_the_whole_image(ebo, synth, config, world, n_windows, stray, flow_amp,
                                                                     halo, expect_escape):
    """SURVEY 8(e)'s band-limited final image: every rank counts its events into its own rows + a halo (LDS tiles, no
    global atomics), the halo rows go to the two neighbours, own + received halos = the rank's rows of the one-process
    image, bit for bit.  The exchange is done here by handing the buffers over in one process (what
    ebo_band_exchange_device's send / recv do between ranks).  Flows beyond the halo raise the flag on a rank that
    could lose events; then the dense path (ebo_count_image_shard + sum) is the answer."""
    import torch
    S = _band_setup(ebo, synth, config, world, n_windows, stray, flow_amp, 23 + config)
    iw, ih, npy, ph = S["iw"], S["ih"], S["npy"], S["ph"]
    bounds = [ebo.shard_range(npy, r, world)[0] * ph for r in range(world)] + [ih]
    if expect_escape is None:
        for r in range(world):
            with pytest.raises(ebo.EboError) as err:
                ebo.band_plan(ih, bounds, r, halo)
            assert err.value.code == ebo.ERR_UNSUPPORTED
        return
    d_flows = torch.from_numpy(S["flows"]).to("cuda")
    ranks = []
    for r in range(world):
        b, e = ebo.shard_range(npy, r, world)
        band = ebo.band_plan(ih, bounds, r, halo)
        assert (band.own_row0, band.own_row1) == (bounds[r], bounds[r + 1])
        R = dict(band=band, ctx=None)
        mk = lambda rows: torch.full((n_windows, rows, iw), -7, dtype=torch.int32, device="cuda") if rows else None
        R["top"], R["own"], R["bottom"] = mk(band.top_rows), mk(band.own_rows), mk(band.bottom_rows)
        R["flag"] = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        if b < e:
            my, sev, soffs = _shard_events(S, b, e, n_windows)
            c = ebo.Context(**S["kw"])
            c.set_stream(torch.cuda.current_stream().cuda_stream)  # the buffers are torch's: one stream, no race with their fills
            c.set_patches(sev, soffs, np.tile(S["rects"][my], (n_windows, 1)))
            ptr = lambda t: t.data_ptr() if t is not None else 0
            c.count_image_band_device(n_windows, S["t_ref"], d_flows.data_ptr(), band, ptr(R["top"]), ptr(R["own"]),
                                      ptr(R["bottom"]), R["flag"].data_ptr())
            c.synchronize()
            R["ctx"] = c
        else:
            R["flag"].zero_()
        ranks.append(R)
    torch.cuda.synchronize()
    escaped = max(int(R["flag"].item()) for R in ranks)  # ebo_band_exchange_device: ncclAllReduce(max)
    assert escaped == (1 if expect_escape else 0)
    if not escaped:
        rows, sent = [], 0
        for r, R in enumerate(ranks):
            band = R["band"]
            above = ranks[r - 1]["bottom"] if r > 0 else None        # what rank r - 1 sends down
            below = ranks[r + 1]["top"] if r + 1 < world else None   # what rank r + 1 sends up
            assert (above.shape[1] if above is not None else 0) == band.recv_above
            assert (below.shape[1] if below is not None else 0) == band.recv_below
            sent = max(sent, (band.top_rows + band.bottom_rows) * iw * 4)
            if R["ctx"] is None:
                continue
            img = torch.zeros((n_windows, band.own_rows, iw), dtype=torch.float64, device="cuda")
            ptr = lambda t: t.data_ptr() if t is not None else 0
            R["ctx"].band_finish_device(n_windows, band, ptr(R["own"]), ptr(above), ptr(below), img.data_ptr())
            R["ctx"].synchronize()
            rows.append(img.cpu().numpy())
        full = np.concatenate(rows, axis=1)
        assert np.array_equal(full, S["whole"])
        assert sent <= 0.25 * ih * iw * 8 or world < 8  # bytes a rank sends per window against the dense f64 image
    else:
        total = np.zeros_like(S["whole"])
        for R in ranks:
            if R["ctx"] is not None:
                total += R["ctx"].count_image_shard(n_windows, S["t_ref"], S["flows"])
        assert np.array_equal(total, S["whole"])
    for R in ranks:
        if R["ctx"] is not None:
            R["ctx"].close()
    assert S["whole"].sum() > 0.5 * sum(len(e) for e in S["evs"])


@pytest.mark.parametrize("seed", range(16))
def test_band_limited_images_on_random_geometries(ebo, seed):
    """The band-limited image against the one-process image on random sensors, grids, shard counts, halos, window
    counts and flows (seeded): whenever no rank raises the flag, own rows + received halos assemble the whole image bit
    for bit; when a rank raises it, an event really could leave its band (the dense sum is still right)."""
    import torch
    rng = np.random.RandomState(1000 + seed)
    iw, ih = int(rng.randint(40, 400)), int(rng.randint(40, 300))
    pw, ph = int(rng.randint(8, 48)), int(rng.randint(8, 40))
    pw, ph = min(pw, iw), min(ph, ih)
    n_windows = int(rng.randint(1, 4))
    n = int(rng.choice([500, 5000, 40000]))
    dur = int(rng.choice([5000, 50000]))
    evs, offs = [], [0]
    for w in range(n_windows):
        t = np.sort(rng.randint(0, dur, n)) + 1_000_000 + w * dur
        x = rng.randint(-4 if seed % 3 == 0 else 0, iw + (4 if seed % 3 == 0 else 0), n)
        y = rng.randint(-4 if seed % 3 == 0 else 0, ih + (4 if seed % 3 == 0 else 0), n)
        evs.append(ebo.make_events(x.astype(np.int32), y.astype(np.int32), t.astype(np.int64), np.where(rng.rand(n) < 0.5, 1, -1).astype(np.int32)))
        offs.append(offs[-1] + n)
    allev = np.concatenate(evs)
    kw = dict(image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=ebo.LOSS_VARIANCE, tv_weight=0.0,
              max_events=len(allev), max_windows=n_windows)
    with ebo.Context(**kw) as c:
        c.set_windows(allev, offs)
        npx, npy, P = c.npx, c.npy, c.P
        amp = float(rng.choice([0.0, 0.3, 1.0, 4.0]))
        flows = rng.uniform(-amp, amp, (n_windows, P, 2))
        whole = c.count_image(ebo.COUNT_WARPED, flows)
        rects = np.array([c.patch_rect(p % npx, p // npx) for p in range(P)])
        t_ref = [c.window_info(w)[0] for w in range(n_windows)]
    world = int(rng.randint(1, min(npy, 8) + 1))
    bounds = [int(rects[ebo.shard_range(npy, r, world)[0] * npx][1]) if ebo.shard_range(npy, r, world)[0] < npy else ih
              for r in range(world)] + [ih]
    bounds[0] = 0
    halo = int(rng.randint(0, 40))
    try:
        bands = [ebo.band_plan(ih, bounds, r, halo) for r in range(world)]
    except ebo.EboError as err:
        assert err.code == ebo.ERR_UNSUPPORTED  # a halo larger than a neighbour: the same verdict for every rank
        return
    S = dict(evs=evs, npx=npx, npy=npy, pw=pw, ph=ph)
    d_flows = torch.from_numpy(flows).to("cuda")
    ranks = []
    for r in range(world):
        b, e = ebo.shard_range(npy, r, world)
        band = bands[r]
        mk = lambda rows: torch.full((n_windows, rows, iw), -3, dtype=torch.int32, device="cuda") if rows else None
        R = dict(band=band, top=mk(band.top_rows), own=mk(band.own_rows), bottom=mk(band.bottom_rows),
                 flag=torch.zeros(1, dtype=torch.int32, device="cuda"), ctx=None)
        if b < e:
            my, sev, soffs = _shard_events(S, b, e, n_windows)
            c = ebo.Context(**kw)
            c.set_stream(torch.cuda.current_stream().cuda_stream)  # the buffers are torch's: one stream, no race with their fills
            c.set_patches(sev, soffs, np.tile(rects[my], (n_windows, 1)))
            ptr = lambda t: t.data_ptr() if t is not None else 0
            c.count_image_band_device(n_windows, t_ref, d_flows.data_ptr(), band, ptr(R["top"]), ptr(R["own"]), ptr(R["bottom"]),
                                      R["flag"].data_ptr())
            c.synchronize()
            R["ctx"] = c
        ranks.append(R)
    torch.cuda.synchronize()
    escaped = max(int(R["flag"].item()) for R in ranks)
    if not escaped:
        rows = []
        for r, R in enumerate(ranks):
            band = R["band"]
            if R["ctx"] is None:
                continue
            above = ranks[r - 1]["bottom"] if r > 0 and band.recv_above else None
            below = ranks[r + 1]["top"] if r + 1 < world and band.recv_below else None
            assert (above.shape[1] if above is not None else 0) == band.recv_above
            assert (below.shape[1] if below is not None else 0) == band.recv_below
            img = torch.zeros((n_windows, band.own_rows, iw), dtype=torch.float64, device="cuda")
            ptr = lambda t: t.data_ptr() if t is not None else 0
            R["ctx"].band_finish_device(n_windows, band, ptr(R["own"]), ptr(above), ptr(below), img.data_ptr())
            R["ctx"].synchronize()
            rows.append(img.cpu().numpy())
        assert np.array_equal(np.concatenate(rows, axis=1), whole)
    else:
        assert amp * dur * 0.5e-3 + 1 > halo or any(R["band"].own_rows == 0 for R in ranks) or seed % 3 == 0
        total = np.zeros_like(whole)
        for R in ranks:
            if R["ctx"] is not None:
                total += R["ctx"].count_image_shard(n_windows, t_ref, flows)
        assert np.array_equal(total, whole)
    for R in ranks:
        if R["ctx"] is not None:
            R["ctx"].close()
