"""CPU tests: the oracle against every known-answer the reference's own tests hold for
this path, and against the probe digits recorded in SURVEY.md §8(c).

Reference tests mirrored here (paths relative to the reference checkout):
  implementation/feature_tracker/test/patch_test.cpp:35-60   Patch.integrateEventsTest
  implementation/feature_tracker/test/patch_test.cpp:7-33    Patch.addEventsTest (window order)
  implementation/feature_tracker/test/feature_detector_test.cpp:43-97  updatePatchTest (membership)
  tools/dataset_reader/test/davis240c_reader_test.cpp:19-48  Davis240cReader.eventsTest
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def make_probe_input(orc):
    """The generating script of the probe input."""
    mask = (1 << 64) - 1
    state = [12345]

    def rnd():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) & mask
        return (state[0] >> 33) & 0xFFFFFFFF

    xs, ys, ts, sg = [], [], [], []
    for k in range(400):
        t = 1000 + 50 * k
        edge = 45 + 0.5 * (t - 1000) * 1e-3
        x = int(edge + (rnd() % 3) - 1)
        y = int(40 + rnd() % 20)
        sign = rnd() & 1
        xs.append(x)
        ys.append(y)
        ts.append(t)
        sg.append(1 if sign else -1)
    return orc.make_events(xs, ys, ts, sg)


def test_probe_reference_time(orc):
    ev = make_probe_input(orc)
    assert orc.mid_timestamp(ev["t_us"][0], ev["t_us"][-1]) == 10975


def test_probe_digits_value_and_jacobian(orc):
    """contrastFunctor value + Jacobian, both losses, vs the SURVEY §8(c) digits."""
    gold = json.load(open(os.path.join(GOLDEN, "survey_probe_contrast.json")))
    ev = make_probe_input(orc)
    rect = gold["input"]["patch_rect"]
    img = orc.contrast_image(ev, rect, (0, 0), 1)
    assert img[0].mean() == pytest.approx(gold["image_mean_at_zero_flow"], rel=1e-13)
    for case in gold["cases"]:
        for loss, key in ((0, "edge"), (1, "var")):
            r, J = orc.contrast_eval(ev, rect, case["m"], loss)
            assert r == pytest.approx(case[key + "_r"], rel=1e-13), (case["m"], key)
            np.testing.assert_allclose(J, case[key + "_J"], rtol=1e-9, atol=1e-16)
            # T=double instantiation == scalar part of the Jet instantiation
            r2, _ = orc.contrast_eval(ev, rect, case["m"], loss, want_jac=False)
            assert r2 == pytest.approx(r, rel=1e-14)


def test_probe_minimum_at_true_flow(orc):
    """Both losses are lowest at the true flow (0.5, 0) among the probe cases."""
    gold = json.load(open(os.path.join(GOLDEN, "survey_probe_contrast.json")))
    for key in ("edge_r", "var_r"):
        best = min(gold["cases"], key=lambda c: c[key])
        assert best["m"] == [0.5, 0.0]


def test_jacobian_matches_finite_differences(orc):
    """The dual-number Jacobian is the derivative of the value where the bin
    assignment does not change (variance loss: smooth between truncation jumps)."""
    ev = make_probe_input(orc)
    rect = (40, 40, 20, 20)
    m = np.array([0.31, -0.17])
    _, J = orc.contrast_eval(ev, rect, m, 1)
    h = 1e-7
    num = np.zeros(2)
    for k in range(2):
        e = np.zeros(2)
        e[k] = h
        rp, _ = orc.contrast_eval(ev, rect, m + e, 1, want_jac=False)
        rm, _ = orc.contrast_eval(ev, rect, m - e, 1, want_jac=False)
        num[k] = (rp - rm) / (2 * h)
    np.testing.assert_allclose(J, num, rtol=2e-5, atol=1e-7)


def test_out_of_window_penalty(orc):
    """contrast_functor.h:143-149 / :159-165: everything warped outside 3x patch."""
    ev = make_probe_input(orc)
    rect = (40, 40, 20, 20)
    m = np.array([1.0e4, -2.0e4])
    for loss in (0, 1):
        r, J = orc.contrast_eval(ev, rect, m, loss)
        assert r == pytest.approx(1e3 * (1 + m[0] ** 2 + m[1] ** 2), rel=1e-15)
        np.testing.assert_allclose(J, 2e3 * m, rtol=1e-15)


# ---- reference known-answer: patch_test.cpp:35-60 -------------------------------
def test_patch_integrate_events_reference_known_answer(orc):
    # Patch({10,10}, extent 3) -> Rect2d(7,7,7,7); 30 events at {7+i/7, 7+i%7},
    # alternating polarity, pushed to the FRONT of the deque (patch.cpp:37-47).
    xs = [7 + i // 7 for i in range(30)]
    ys = [7 + i % 7 for i in range(30)]
    sg = [1 if i % 2 == 0 else -1 for i in range(30)]
    ts = list(range(30))
    ev = orc.make_events(xs, ys, ts, sg)[::-1].copy()  # deque order: newest first
    nabla, cur, last = orc.patch_integrate(ev, (7.0, 7.0, 7.0, 7.0))
    for i in range(30):
        assert nabla[i % 7, i // 7] == (1.0 if i % 2 == 0 else -1.0)
    assert nabla.sum() == 0.0
    assert np.count_nonzero(nabla) == 30
    # patch.cpp:78-83: mid of newest/oldest, and the OLDEST event's time
    assert cur == 14
    assert last == 0


def test_patch_window_order_reference(orc):
    # patch_test.cpp:24-28: front of the window is the first event pushed when
    # read through getEvents() ... the deque holds newest first; the test's
    # front().timestamp == 0 documents that events_.front() is the LAST pushed
    # only after pop: with 30 events and numOfEvents_=max(100,30) nothing pops.
    ev = orc.make_events([8] * 3, [8] * 3, [5, 6, 7], [1, 1, 1])[::-1].copy()
    _, cur, last = orc.patch_integrate(ev, (7.0, 7.0, 7.0, 7.0))
    assert cur == 6 and last == 5


def test_patch_rect2d_membership_half_open(orc):
    # cv::Rect2d::contains: x <= px < x+w (feature_detector_test.cpp:43-97 relies on it)
    ev = orc.make_events([6, 7, 13, 14, 7], [7, 7, 13, 13, 14], [0, 1, 2, 3, 4], [1] * 5)
    nabla, _, _ = orc.patch_integrate(ev, (7.0, 7.0, 7.0, 7.0))
    assert nabla[0, 0] == 1.0 and nabla[6, 6] == 1.0
    assert nabla.sum() == 2.0
    # non-integer rect (after Patch::updatePatchRect): int - double truncates
    nabla, _, _ = orc.patch_integrate(ev, (6.5, 6.5, 7.0, 7.0))
    assert nabla[0, 0] == 1.0  # (7,7) -> (0.5,0.5) -> (0,0)
    assert nabla[6, 6] == 1.0  # (13,13) -> (6.5,6.5) -> (6,6)
    assert nabla.sum() == 2.0


def test_update_patches_routing_reference_scenario(orc):
    """feature_detector_test.cpp:43-97 (updatePatchTest): patches of extent 11 at (0,0), (5,5),
    (20,20), events in [0,30)^2; a patch's events are those a plain isInPatch test selects, in
    stream order.  Also the quota / start / next bookkeeping the batched caller relies on."""
    rng = np.random.RandomState(3)
    n = 400
    x, y = rng.randint(0, 30, n), rng.randint(0, 30, n)
    ev = orc.make_events(x, y, np.arange(n), np.where(rng.rand(n) < 0.5, 1, -1))
    rects = np.array([[c - 11.0, c - 11.0, 23.0, 23.0] for c in (0.0, 5.0, 20.0)])
    inside = [np.flatnonzero((rects[p, 0] <= x) & (x < rects[p, 0] + 23) & (rects[p, 1] <= y) & (y < rects[p, 1] + 23))
              for p in range(3)]
    got, nxt = orc.route_events(ev, rects, [0, 0, 0], [10**6] * 3, n)
    for p in range(3):
        assert np.array_equal(got[p], inside[p])
    assert list(nxt) == [n, n, n]
    # quota 30 from a late start: the 30 first members at or after `start`, next = after the 30th
    got, nxt = orc.route_events(ev, rects, [50, 0, 399], [30, 0, 5], 64)
    late = inside[0][inside[0] >= 50]
    assert np.array_equal(got[0], late[:30]) and nxt[0] == late[29] + 1
    assert len(got[1]) == 0 and nxt[1] == 0          # quota 0: nothing taken, nothing skipped
    assert nxt[2] == n                                # chunk exhausted before the quota


def test_patch_integrate_mc_round_half_even(orc):
    # patch.cpp:118-119 Point2d -> Point2i is cvRound (half to even), SURVEY F7.
    # one event at x=10, flow so that compensated x = 10.5 and 11.5
    ev = orc.make_events([10, 11], [10, 10], [0, 0], [1, 1])
    # dir = (1, 0) over t_dif = 2 -> (t - t_e)/t_dif * dir = 0.5 at t = 1
    nabla, upd = orc.patch_integrate_mc(ev, (5.0, 5.0, 11.0, 11.0), (0.0, 0.0, 0), (1.0, 0.0, 2), 1)
    assert upd
    assert nabla[5, 5] == 1.0  # 10.5 -> 10
    assert nabla[5, 7] == 1.0  # 11.5 -> 12
    assert nabla.sum() == 2.0
    # time test fails (patch.cpp:99-100): image untouched
    nabla, upd = orc.patch_integrate_mc(ev, (5.0, 5.0, 11.0, 11.0), (0.0, 0.0, 0), (1.0, 0.0, 2), 10)
    assert not upd


# ---- reference fixture: davis240c_reader_test.cpp:19-48 --------------------------
def test_davis_event_text_fixture(orc):
    rc, ev = orc.parse_events_txt(os.path.join(GOLDEN, "davis_events_fixture.txt"))
    assert rc == 0
    assert len(ev) == 5
    assert ev["x"].tolist() == [33, 158, 88, 174, 112]
    assert ev["y"].tolist() == [39, 145, 143, 154, 139]
    assert ev["sign"].tolist() == [1, 1, -1, -1, 1]
    assert ev["t_us"].tolist() == [0, 11, 50, 55, 80]


def test_davis_event_text_bad_sign(orc, tmp_path):
    p = tmp_path / "events.txt"
    p.write_text("0.000001 1 2 1\n0.000002 3 4 2\n")
    rc, ev = orc.parse_events_txt(str(p))
    assert rc == -3  # "Sign is not equal to 0/1" (davis240c_reader.cpp:85-88)
    assert len(ev) == 1


# ---- grid, bucketing, count images ----------------------------------------------
def test_grid_remainder_rule(orc):
    prm = orc.default_params(image_w=240, image_h=180, patch_w=30, patch_h=22)
    assert orc.grid(prm) == (8, 8)
    assert orc.patch_rect(prm, 0, 0) == (0, 0, 30, 22)
    assert orc.patch_rect(prm, 7, 7) == (210, 154, 30, 26)  # last row absorbs 180-8*22
    prm = orc.default_params(image_w=346, image_h=260, patch_w=21, patch_h=16)
    assert orc.grid(prm) == (16, 16)
    assert orc.patch_rect(prm, 15, 15) == (315, 240, 31, 20)


def test_integrate_and_final_image(orc, synth):
    ev, _ = synth.make_window(0, n_events=3000)
    img = orc.integrate_events(ev, 240, 180)
    assert img.sum() == 3000
    ref = np.zeros((180, 240))
    np.add.at(ref, (ev["y"], ev["x"]), 1.0)
    assert np.array_equal(img, ref)
    prm = orc.default_params()
    npx, npy = orc.grid(prm)
    # zero flow: warped == un-warped
    img0 = orc.final_count_image(ev, prm, np.zeros((npx * npy, 2)))
    assert np.array_equal(img0, ref)
    # out-of-sensor events are dropped by the bounds check
    ev2 = ev.copy()
    ev2["x"][:10] = -5
    assert orc.integrate_events(ev2, 240, 180).sum() == 2990


def test_final_image_round_half_away(orc):
    # feature_detector.cpp:446-453 round(): -0.5 -> -1 (dropped), 2.5 -> 3
    prm = orc.default_params(image_w=40, image_h=40, patch_w=20, patch_h=20)
    ev = orc.make_events([0, 2, 5], [0, 2, 5], [0, 1000, 2000], [1, 1, 1])
    # t_ref = 1000; dt*scale = (1, 0, -1) ms
    flows = np.zeros((4, 2))
    flows[0] = (-0.5, 0.5)
    img = orc.final_count_image(ev, prm, flows)
    # event0: x = 0 - 0.5 -> round(-0.5) = -1 -> dropped
    # event1: dt = 0 -> (2,2); event2: x = 5 + 0.5 = 5.5 -> 6, y = 5 - 0.5 = 4.5 -> 5
    assert img[2, 2] == 1 and img[5, 6] == 1 and img.sum() == 2


def test_field_image_uses_float32_field(orc):
    ev = orc.make_events([10, 10], [10, 10], [0, 2000], [1, 1])
    field = np.zeros((20, 20, 2), dtype=np.float32)
    field[10, 10] = (1.5, 0.0)
    img = orc.compensate_events_field(ev, 20, 20, field)
    # t_ref = 1000: first event +1 ms * 1.5 = 11.5 -> 12, second -1.5 -> 8.5 -> 9
    assert img[10, 12] == 1 and img[10, 9] == 1


# ---- TV functor and solver ---------------------------------------------------------
def test_total_variance_functor(orc):
    r, jx, jy = orc.tv_eval(1e3, (0.5, -0.25), (0.25, 0.0))
    np.testing.assert_allclose(r, [250.0, 250.0])
    np.testing.assert_allclose(jx, [[1e3, 0], [0, -1e3]])
    np.testing.assert_allclose(jy, [[-1e3, 0], [0, 1e3]])
    # ceres::abs at 0 keeps +f (total_variance.h:17-18)
    r, jx, jy = orc.tv_eval(2.0, (0.0, 0.0), (0.0, 0.0))
    np.testing.assert_allclose(r, [0, 0])
    np.testing.assert_allclose(jx, [[2, 0], [0, 2]])


def _edge_window(orc, flow=(0.4, -0.2), n=1500, seed=7):
    """Events of one straight edge moving at `flow` px/ms across a 60x60 sensor."""
    rng = np.random.RandomState(seed)
    t = np.sort(rng.randint(0, 30000, n)).astype(np.int64) + 100000
    s = rng.uniform(-12, 12, n)
    dt = (t - 115000) * 1e-3
    x = 30 + s * 0.3 + flow[0] * dt + rng.randint(-1, 2, n)
    y = 30 + s * 0.95 + flow[1] * dt + rng.randint(-1, 2, n)
    keep = (x >= 0) & (x < 60) & (y >= 0) & (y < 60)
    return orc.make_events(np.floor(x[keep]).astype(int), np.floor(y[keep]).astype(int), t[keep])


def test_solver_reduces_cost_and_modes_agree_on_single_patch(orc):
    ev = _edge_window(orc)
    prm = orc.default_params(image_w=60, image_h=60, patch_w=60, patch_h=60, loss=1, tv_weight=0.0)
    o0 = orc.default_solver(mode=0)
    f0, img, s0 = orc.compensate_events_contrast(ev, prm, o0)
    o1 = orc.default_solver(mode=1)
    f1, _, s1 = orc.compensate_events_contrast(ev, prm, o1)
    # one patch, no TV: the global problem IS the per-patch problem
    np.testing.assert_array_equal(f0, f1)
    assert s0.final_cost < s0.initial_cost
    assert s0.num_evals_jac >= 1 and s0.iterations <= 50
    assert img.sum() <= len(ev)
    # the objective at the solution is not worse than at zero flow
    r_sol, _, _, _ = orc.window_eval(ev, prm, f0, want_jac=False)
    r_zero, _, _, _ = orc.window_eval(ev, prm, np.zeros((1, 2)), want_jac=False)
    assert r_sol[0] <= r_zero[0]


def test_solver_tv_couples_patches(orc, synth):
    ev, _ = synth.make_window(0, n_events=6000)
    prm = orc.default_params(loss=1)  # reference defaults: 12x9 patches, TV 1e3
    f_tv, _, s = orc.compensate_events_contrast(ev, prm, orc.default_solver(max_num_iterations=8), want_image=False)
    prm0 = orc.default_params(loss=1, tv_weight=0.0)
    f_no, _, _ = orc.compensate_events_contrast(ev, prm0, orc.default_solver(max_num_iterations=8), want_image=False)
    r, _, active, counts = orc.window_eval(ev, prm, np.zeros((108, 2)), want_jac=False)
    assert counts.sum() == 6000
    # patches without a data term stay at zero without TV ...
    assert np.all(f_no[active == 0] == 0.0)
    # ... and the two problems are different problems
    assert s.iterations <= 8
    assert f_tv.shape == (108, 2)


def test_tv_free_solve_is_chaotic_on_the_cpu_alone(orc, synth):
    """Evidence for how solver parity is tested on the GPU: without the TV terms the
    reference's own formulation is chaotic.  Perturbing compensateScale by ONE ulp
    leaves the CPU path's answer unchanged to ~1e-9 for the first ~20 LM iterations
    and moves it by more than 1e-2 px/ms after 50.  With the reference's TV coupling
    (its default) the same perturbation changes nothing measurable."""
    ev, _ = synth.make_window(0, n_events=15000)

    def solve(tv, scale, iters, mode):
        prm = orc.default_params(loss=1, tv_weight=tv, scale=scale)
        f, _, _ = orc.compensate_events_contrast(
            ev, prm, orc.default_solver(mode=mode, max_num_iterations=iters), want_image=False)
        return f

    s0, s1 = 1e-3, float(np.nextafter(1e-3, 1.0))
    d20 = np.abs(solve(0.0, s0, 20, 1) - solve(0.0, s1, 20, 1)).max()
    d50 = np.abs(solve(0.0, s0, 50, 1) - solve(0.0, s1, 50, 1)).max()
    assert d20 < 1e-7
    assert d50 > 1e-2
    tv50 = np.abs(solve(1e3, s0, 50, 0) - solve(1e3, s1, 50, 0)).max()
    assert tv50 < 1e-9


def test_lm_restatement_reproduces_the_table_ceres_publishes(orc):
    """The trust-region LM the reference delegates to ceres::Solve, as restated in
    oracle.cpp::minimize (Jacobi scaling 1 / (1 + |column|), LM diagonal clamped and divided by
    the radius, step quality, radius update, termination tests), run on Powell's function exactly
    as Ceres' tutorial sets it up, prints the table that tutorial publishes -- every column, every
    digit, all 14 iterations, the termination reason and the final point."""
    import json
    gold = json.load(open(os.path.join(HERE, "golden", "ceres_tutorial_powell.json")))
    x, trace, s = orc.lm_powell(gold["initial_x"])
    rows = gold["rows"]
    assert "%.6e" % s.initial_cost == rows[0][1]
    assert len(trace) == len(rows) - 1 == s.iterations
    assert s.termination == 0  # CONVERGENCE, by the gradient tolerance (3.64e-11 <= 1e-10)
    for got, want in zip(trace, rows[1:]):
        printed = ["%.6e" % got[0]] + ["%.2e" % v for v in got[1:]]
        assert printed == want[1:], (want[0], printed, want[1:])
    assert ["%g" % v for v in x] == gold["final_x_as_printed"]
    assert "%.6e" % s.final_cost == rows[-1][1]
