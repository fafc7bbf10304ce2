"""GPU tests of ebo_set_patches: arbitrary cv::Rect2i patches with their own event
lists, i.e. exactly what tracker::contrastFunctor's constructor takes
(contrast_functor.h:12-21).  Includes the SURVEY §8(c) probe evaluated by the HIP
kernels against the recorded digits."""
import json
import os

import numpy as np
import pytest

from test_oracle_golden import GOLDEN, make_probe_input

pytestmark = pytest.mark.gpu


def test_probe_digits_through_the_hip_path(ebo, orc):
    gold = json.load(open(os.path.join(GOLDEN, "survey_probe_contrast.json")))
    ev = make_probe_input(orc)
    rect = gold["input"]["patch_rect"]
    with ebo.Context(loss=ebo.LOSS_VARIANCE, min_events=100) as c:
        c.set_patches(ev, [0, len(ev)], [rect])
        assert c.patch_info(0) == (400, True, gold["input"]["timestamp_ref"])
        for case in gold["cases"]:
            r, J = c.eval([case["m"]])
            assert r[0, 0] == pytest.approx(case["var_r"], rel=1e-12)
            np.testing.assert_allclose(J[0, 0], case["var_J"], rtol=1e-9, atol=1e-14)
            r1, _ = c.eval([case["m"]], want_jac=False)
            assert r1[0, 0] == pytest.approx(case["var_r"], rel=1e-12)
        img = c.contrast_image(0, (0.0, 0.0), 1)
        assert img[0].mean() == pytest.approx(gold["image_mean_at_zero_flow"], rel=1e-12)


def test_arbitrary_rects_and_unfiltered_events(ebo, orc):
    """Overlapping / off-grid rects; events outside their rect are still splatted when
    they warp into the 3x window (the functor never filters by the rect)."""
    rng = np.random.RandomState(21)
    rects, evs, offs = [], [], [0]
    for k in range(12):
        w, h = rng.randint(8, 40), rng.randint(8, 40)
        x, y = rng.randint(0, 200), rng.randint(0, 140)
        n = rng.randint(101, 900)
        ex = rng.randint(x - w // 2, x + w + w // 2, n)
        ey = rng.randint(y - h // 2, y + h + h // 2, n)
        t = np.sort(rng.randint(0, 40000, n)) + 7000 * k + 1
        evs.append(orc.make_events(ex, ey, t))
        offs.append(offs[-1] + n)
        rects.append((x, y, w, h))
    ev = np.concatenate(evs)
    flows = rng.uniform(-1.2, 1.2, (12, 2))
    with ebo.Context(loss=ebo.LOSS_VARIANCE, max_events=len(ev)) as c:
        c.set_patches(ev, offs, rects)
        r, J = c.eval(flows)
        for k in range(12):
            ro, Jo = orc.contrast_eval(evs[k], rects[k], flows[k], 1)
            assert r[0, k] == pytest.approx(ro, rel=1e-9)
            np.testing.assert_allclose(J[0, k], Jo, rtol=1e-9, atol=1e-10)
            img = c.contrast_image(k, flows[k], 3)
            np.testing.assert_allclose(img, orc.contrast_image(evs[k], rects[k], flows[k], 3),
                                       rtol=1e-11, atol=1e-13)
        # per-patch solve on the device for arbitrary patches (capped: chaos horizon)
        sol, summ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=12)
        assert sol.shape == (1, 12, 2) and np.isfinite(sol).all()
        with pytest.raises(ebo.EboError) as ei:
            c.count_image(ebo.COUNT_INTEGRATED)
        assert ei.value.code == ebo.ERR_STATE
        with pytest.raises(ebo.EboError) as ei:
            c.solve(mode=ebo.SOLVE_GLOBAL)
        assert ei.value.code == ebo.ERR_UNSUPPORTED
        # back to a window: grid geometry is restored
        c.set_window(ev)
        r, _ = c.eval(np.zeros((c.P, 2)))
        assert r.shape == (1, c.P)


def test_negative_coordinates_truncate_toward_zero(ebo, orc):
    """int(c) truncates toward zero (contrast_functor.h:59): warped coordinates in
    (-1, 0) bin to 0, not -1.  Rect placed at the origin, flows push events negative."""
    rng = np.random.RandomState(2)
    n = 600
    ev = orc.make_events(rng.randint(0, 6, n), rng.randint(0, 6, n), np.sort(rng.randint(0, 20000, n)) + 5)
    rect = (0, 0, 12, 12)
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        c.set_patches(ev, [0, n], [rect])
        for m in ((0.37, -0.41), (-0.9, 0.9), (0.05, 0.05)):
            r, J = c.eval([m])
            ro, Jo = orc.contrast_eval(ev, rect, m, 1)
            assert r[0, 0] == pytest.approx(ro, rel=1e-9)
            np.testing.assert_allclose(J[0, 0], Jo, rtol=1e-9, atol=1e-10)
