"""Event -> tracked-patch routing (FeatureDetector::updatePatches, feature_detector.cpp:585-596)
on the device against the oracle's restatement of the reference loop: same indices, in stream
order, same stop position; fractional and overlapping rects, quotas, late starts, empty cases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def chunk(ebo, rng, n, w=240, h=180):
    x = rng.randint(-4, w + 4, n).astype(np.int32)
    y = rng.randint(-4, h + 4, n).astype(np.int32)
    t = np.sort(rng.randint(0, 200000, n)) + 1000
    return ebo.make_events(x, y, t, np.where(rng.rand(n) < 0.5, 1, -1))


@pytest.mark.parametrize("n_events,n_patches", [(1, 3), (63, 5), (64, 5), (257, 40), (20000, 100), (120000, 300)])
def test_routing_matches_the_reference_loop(ebo, orc, n_events, n_patches):
    rng = np.random.RandomState(n_events + n_patches)
    ev = chunk(ebo, rng, n_events)
    ext = rng.randint(2, 16, n_patches)
    cx, cy = rng.uniform(-5, 245, n_patches), rng.uniform(-5, 185, n_patches)
    frac = rng.choice([0.0, 0.5, 0.25], n_patches)
    rects = np.stack([cx - ext + frac, cy - ext - frac, 2 * ext + 1.0, 2 * ext + 1.0], 1)
    rects[0] = rects[min(1, n_patches - 1)]  # two patches on the same pixels: an event goes to both
    start = rng.randint(0, n_events + 2, n_patches).astype(np.uint32)
    start[: n_patches // 2] = 0
    take = rng.choice([0, 1, 30, 75, 300, 10**6], n_patches).astype(np.uint32)
    cap = 300
    with ebo.Context(image_w=240, image_h=180) as c:
        c.route_set_events(ev)
        got, nxt = c.route_events(rects, start, take, cap)
        want, wnxt = orc.route_events(ev, rects, start, take, cap)
        for p in range(n_patches):
            assert np.array_equal(got[p], want[p]), p
        assert np.array_equal(nxt, wnxt)
        # a second call on the same chunk from where the first stopped (rects moved)
        rects2 = rects + np.array([0.75, -1.5, 0.0, 0.0])
        got2, nxt2 = c.route_events(rects2, nxt, take, cap)
        want2, wnxt2 = orc.route_events(ev, rects2, wnxt, take, cap)
        for p in range(n_patches):
            assert np.array_equal(got2[p], want2[p]), p
        assert np.array_equal(nxt2, wnxt2)


def test_routing_edge_cases(ebo, orc):
    with ebo.Context(image_w=240, image_h=180) as c:
        c.route_set_events(np.zeros(0, dtype=ebo.EVENT_DTYPE))
        got, nxt = c.route_events([[0, 0, 5, 5]], [0], [10], 16)
        assert len(got[0]) == 0 and nxt[0] == 0
        rng = np.random.RandomState(1)
        ev = chunk(ebo, rng, 500)
        c.route_set_events(ev)
        # the quota is filled by the very last event of the chunk: next == n
        inside = np.flatnonzero((ev["x"] >= 10) & (ev["x"] < 200) & (ev["y"] >= 10) & (ev["y"] < 150))
        got, nxt = c.route_events([[10, 10, 190, 140]], [0], [len(inside)], 1000)
        assert np.array_equal(got[0], inside) and nxt[0] == inside[-1] + 1
        # cap below the quota: the list stops at cap
        got, nxt = c.route_events([[10, 10, 190, 140]], [0], [10**6], 7)
        assert np.array_equal(got[0], inside[:7]) and nxt[0] == inside[6] + 1
    with ebo.Context(image_w=240, image_h=180) as c:
        bad = ebo.make_events([20000], [5], [1], [1])
        with pytest.raises(ebo.EboError):
            c.route_set_events(bad)
