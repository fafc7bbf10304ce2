"""Synthetic event windows for tests and bench.py.

Counter-based SplitMix64 so that any language reproduces the same stream:
``u64[i] = mix(seed + (i+1) * 0x9E3779B97F4A7C15)``.  seed = 20200701 + config index
(+ 1000 * window index for further windows of a stream).

Per patch p a ground-truth flow v_p ~ U(-vmax, vmax)^2 px/ms; 90 % of the events
drawn for that patch lie on one of its 1-3 straight edges translating at v_p
(edge point at the event's time, +-1 px jitter), 10 % are uniform noise inside the
patch.  Timestamps are uniform over the 50 ms window and sorted; polarity is a
fair coin; coordinates are clamped into the sensor.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
WINDOW_US = 50_000

# BASELINE.json configs: sensor, patch size giving the named patch count with the
# reference's grid rule (feature_detector.cpp:301-346), events per window.
CONFIGS = {
    1: dict(name="C1 240x180 1 patch 10k ev", image=(240, 180), patch=(240, 180), events=10_000),
    2: dict(name="C2 240x180 64 patches 50k ev", image=(240, 180), patch=(30, 22), events=50_000),
    3: dict(name="C3 346x260 256 patches 200k ev", image=(346, 260), patch=(21, 16), events=200_000),
    4: dict(name="C4 1280x720 1024 patches 1M ev", image=(1280, 720), patch=(40, 22), events=1_000_000),
    # the reference's own defaults (DetectorParams): 12x9 patches of 20x20, 15k-event window
    0: dict(name="reference defaults 240x180 108 patches 15k ev", image=(240, 180), patch=(20, 20), events=15_000),
}


def splitmix64(seed, n, start=0):
    """n outputs of the counter-based SplitMix64 stream, starting at counter `start`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _unit(u64):
    """uint64 -> double in [0,1) with 53 random bits."""
    return (u64 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def grid_rects(image, patch):
    """Patch rects (x, y, w, h) in row-major grid order, feature_detector.cpp:301-346."""
    iw, ih = image
    pw, ph = patch
    npx, npy = iw // pw, ih // ph
    rects = []
    for y in range(npy):
        for x in range(npx):
            w = pw if x < npx - 1 else iw - x * pw
            h = ph if y < npy - 1 else ih - y * ph
            rects.append((x * pw, y * ph, w, h))
    return npx, npy, np.array(rects, dtype=np.int64)


def make_window(config, window=0, n_events=None, vmax=1.0, t0_us=1_000_000, event_dtype=None):
    """Returns (events structured array sorted by time, ground-truth flows [P][2])."""
    cfg = CONFIGS[config] if not isinstance(config, dict) else config
    idx = config if not isinstance(config, dict) else cfg.get("index", 9)
    seed = 20200701 + idx + 1000 * window
    iw, ih = cfg["image"]
    npx, npy, rects = grid_rects(cfg["image"], cfg["patch"])
    P = npx * npy
    N = int(n_events if n_events is not None else cfg["events"])

    # per-patch draws: 2 (flow) + 1 (edge count) + 3 edges * 4
    pp = _unit(splitmix64(seed, P * 15)).reshape(P, 15)
    flow = (pp[:, 0:2] * 2.0 - 1.0) * vmax
    n_edges = 1 + (pp[:, 2] * 3.0).astype(np.int64)
    rx, ry, rw, rh = [rects[:, k].astype(np.float64) for k in range(4)]
    ex = rx[:, None] + pp[:, 3:6] * rw[:, None]
    ey = ry[:, None] + pp[:, 6:9] * rh[:, None]
    eth = pp[:, 9:12] * np.pi
    elen = (0.25 + 0.25 * pp[:, 12:15]) * np.minimum(rw, rh)[:, None]

    # per-event draws: 8 each
    ee = _unit(splitmix64(seed, N * 8, start=P * 15)).reshape(N, 8)
    t = np.sort((ee[:, 0] * WINDOW_US).astype(np.int64), kind="stable")
    pid = np.minimum((ee[:, 1] * P).astype(np.int64), P - 1)
    noise = ee[:, 2] < 0.1
    eidx = np.minimum((ee[:, 3] * n_edges[pid]).astype(np.int64), n_edges[pid] - 1)
    s = ee[:, 4] * 2.0 - 1.0
    jx = np.floor(ee[:, 5] * 3.0) - 1.0
    jy = np.floor(ee[:, 6] * 3.0) - 1.0
    pol = np.where(ee[:, 7] < 0.5, -1, 1)

    dt_ms = (t.astype(np.float64) - WINDOW_US / 2.0) * 1e-3
    th = eth[pid, eidx]
    px = ex[pid, eidx] + s * elen[pid, eidx] * np.cos(th) + flow[pid, 0] * dt_ms + jx
    py = ey[pid, eidx] + s * elen[pid, eidx] * np.sin(th) + flow[pid, 1] * dt_ms + jy
    nx = rx[pid] + ee[:, 4] * rw[pid]
    ny = ry[pid] + ee[:, 5] * rh[pid]
    x = np.where(noise, nx, px)
    y = np.where(noise, ny, py)
    xi = np.clip(np.floor(x), 0, iw - 1).astype(np.int32)
    yi = np.clip(np.floor(y), 0, ih - 1).astype(np.int32)

    if event_dtype is None:
        event_dtype = np.dtype(
            [("x", "<i4"), ("y", "<i4"), ("sign", "<i4"), ("reserved", "<i4"), ("t_us", "<i8")])
    ev = np.zeros(N, dtype=event_dtype)
    ev["x"] = xi
    ev["y"] = yi
    ev["sign"] = pol
    ev["t_us"] = t + t0_us + window * WINDOW_US
    return ev, flow


def make_stream(config, n_windows, **kw):
    """n_windows consecutive windows: (events, offsets [n+1], flows [n][P][2])."""
    evs, flows = [], []
    for w in range(n_windows):
        e, f = make_window(config, window=w, **kw)
        evs.append(e)
        flows.append(f)
    offsets = np.zeros(n_windows + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    return np.concatenate(evs), offsets, np.stack(flows)


def write_events_txt(path, ev):
    """DAVIS events.txt format (davis240c_reader.cpp:60-92): '<sec> <x> <y> <0|1>'."""
    with open(path, "w") as fp:
        for e in ev:
            fp.write("%.9f %d %d %d\n" % (int(e["t_us"]) * 1e-6, e["x"], e["y"], 1 if e["sign"] > 0 else 0))
