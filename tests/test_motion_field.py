"""FeatureDetector::initMotionField (feature_detector.cpp:53-142) and the compensateEvents path
built on it (:243-296).  CPU: the oracle's semantics on hand-checkable cases; GPU: bit-exact
parity of the field (float32) and of the resulting count image."""
import numpy as np
import pytest


def make_trajectories(rng, n, w=240, h=180):
    trajs = []
    for k in range(n):
        m = rng.randint(1, 8)
        t0 = rng.randint(1000, 50000)
        x, y = rng.uniform(5, w - 5), rng.uniform(5, h - 5)
        tr = []
        for i in range(m):
            tr.append((x, y, t0))
            x += rng.uniform(-4, 4)
            y += rng.uniform(-4, 4)
            t0 += rng.randint(2000, 20000)
        trajs.append(tr)
    return trajs


def test_init_motion_field_semantics(orc):
    w, h = 20, 12
    # one patch: segment (5.4, 3.6) -> (7.4, 2.6) over 4000 us, queried inside the first segment
    tr = [[(5.4, 3.6, 1000), (7.4, 2.6, 5000), (9.0, 2.0, 9000)]]
    field, fixed = orc.init_motion_field(w, h, 800, tr, use_average=True)
    # lower_bound(800) -> first sample; velocity = (1/1e-3) * d / dt  px per ms
    assert fixed.tolist() == [[5, 4]]  # round(5.4), round(3.6)
    assert field[4, 5, 0] == np.float32(1000.0 * 2.0 / 4000.0)
    assert field[4, 5, 1] == np.float32(1000.0 * -1.0 / 4000.0)
    assert np.all(field[..., 0] == field[4, 5, 0])  # average of one point everywhere
    # timestamp after the last-but-one sample: low+1 == end -> nothing fixed -> all zero
    field, fixed = orc.init_motion_field(w, h, 9000, tr)
    assert len(fixed) == 0 and not field.any()
    # timestamp 0: "timestamp.count() > 0" fails
    field, fixed = orc.init_motion_field(w, h, 0, tr)
    assert len(fixed) == 0 and not field.any()
    # nearest fill: two fixed points, left/right halves
    tr2 = [[(2.0, 6.0, 100), (3.0, 6.0, 1100)], [(17.0, 6.0, 100), (17.0, 8.0, 1100)]]
    field, fixed = orc.init_motion_field(w, h, 50, tr2, use_average=False)
    assert fixed.tolist() == [[2, 6], [17, 6]]
    assert field[0, 0, 0] == np.float32(1.0) and field[0, 0, 1] == 0
    assert field[11, 19, 0] == 0 and field[11, 19, 1] == np.float32(2.0)
    assert field[6, 9, 0] == np.float32(1.0)   # distance 7 vs 8
    assert field[6, 10, 1] == np.float32(2.0)  # 8 vs 7
    # tie: the first fixed point in list order wins (strict <)
    tr3 = [[(4.0, 6.0, 100), (5.0, 6.0, 1100)], [(14.0, 6.0, 100), (14.0, 8.0, 1100)]]
    field, _ = orc.init_motion_field(w, h, 50, tr3, use_average=False)
    assert field[6, 9, 0] == np.float32(1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("use_average", [True, False])
def test_init_motion_field_on_device_bit_exact(ebo, orc, synth, use_average):
    rng = np.random.RandomState(8)
    trajs = make_trajectories(rng, 60)
    trajs.append([(239.6, 100.0, 2000), (239.9, 101.0, 30000)])  # rounds to x = 240: outside -> skipped
    trajs.append([(50.2, 60.7, 2000), (50.2, 60.7, 30000)])       # zero velocity fixed point
    ev, _ = synth.make_window(0, n_events=15000)
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        for ts in (1, 15000, 30000, 10 ** 7):
            field, fixed = c.init_motion_field(ts, trajs, use_average)
            fo, fxo = orc.init_motion_field(240, 180, ts, trajs, use_average)
            assert np.array_equal(fixed, fxo)
            assert np.array_equal(field.view(np.uint32), fo.view(np.uint32))  # bit exact float32
        field, _ = c.init_motion_field(15000, trajs, use_average)
        # compensateEvents end to end: field stays on the device, warp + count
        c.set_window(ev)
        img = c.count_image(ebo.COUNT_FIELD, None)[0]
        assert np.array_equal(img, orc.compensate_events_field(ev, 240, 180, field))
        assert np.array_equal(img, c.count_image(ebo.COUNT_FIELD, field)[0])
