"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bars:
  * integer count images: bit-exact (np.array_equal)
  * objective value / Jacobian: relative 1e-9 (summation order differs)
  * solved flow: max abs difference <= 1e-5 px/ms against the oracle running the
    same trust-region algorithm
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def oparams(orc, p):
    return orc.default_params(
        image_w=p.image_w, image_h=p.image_h, patch_w=p.patch_w, patch_h=p.patch_h,
        tv_weight=p.tv_weight, tv_huber=p.tv_huber, scale=p.scale, min_events=p.min_events,
        loss=p.loss)


def ctx_for(ebo, synth, config, **kw):
    cfg = synth.CONFIGS[config]
    args = dict(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE)
    args.update(kw)
    return ebo.Context(**args)


def assert_close(a, b, rtol=RTOL, atol=1e-12):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("config,n_events", [(0, 15000), (2, 50000), (3, 40000), (4, 300000)])
def test_eval_value_and_jacobian(ebo, orc, synth, config, n_events):
    ev, gt = synth.make_window(config, n_events=n_events)
    with ctx_for(ebo, synth, config) as c:
        c.set_window(ev)
        rng = np.random.RandomState(config)
        for flows in (np.zeros((c.P, 2)), gt * 0.5, rng.uniform(-1, 1, (c.P, 2))):
            r, J = c.eval(flows)
            ro, Jo, active, counts = orc.window_eval(ev, oparams(orc, c.params), flows)
            for p in range(c.P):
                n, a, _ = c.patch_info(p)
                assert n == counts[p] and a == bool(active[p])
            assert_close(r[0], ro)
            assert_close(J[0], Jo, atol=1e-10)
            r1, _ = c.eval(flows, want_jac=False)
            assert_close(r1[0], ro)
            assert np.all(r[0][active == 0] == 0.0)


def test_eval_reference_times(ebo, orc, synth):
    """Window and per-patch reference times (feature_detector.cpp:305-306,
    contrast_functor.h:18-20) are the int32-truncated mid times."""
    ev, _ = synth.make_window(2, n_events=20000)
    with ctx_for(ebo, synth, 2) as c:
        c.set_window(ev)
        t, n = c.window_info()
        assert n == len(ev)
        assert t == orc.mid_timestamp(ev["t_us"][0], ev["t_us"][-1])
        for p in (0, 9, 63):
            x, y, w, h = c.patch_rect(p % c.npx, p // c.npx)
            m = (ev["x"] >= x) & (ev["x"] < x + w) & (ev["y"] >= y) & (ev["y"] < y + h)
            sub = ev[m]
            cnt, _, tp = c.patch_info(p)
            assert cnt == len(sub)
            assert tp == orc.mid_timestamp(sub["t_us"][0], sub["t_us"][-1])


def test_contrast_image_channels(ebo, orc, synth):
    """The image of warped events itself (contrast_functor.h:38-88), all 3 Jet channels."""
    ev, gt = synth.make_window(2, n_events=30000)
    with ctx_for(ebo, synth, 2) as c:
        c.set_window(ev)
        for p in (0, 27, 63):  # interior, interior, remainder corner (30x26)
            rect = c.patch_rect(p % c.npx, p // c.npx)
            x, y, w, h = rect
            m = (ev["x"] >= x) & (ev["x"] < x + w) & (ev["y"] >= y) & (ev["y"] < y + h)
            for flow in ((0.0, 0.0), tuple(gt[p]), (-0.83, 0.41)):
                img = c.contrast_image(p, flow, 3)
                ref = orc.contrast_image(ev[m], rect, flow, 3)
                assert img.shape == ref.shape
                np.testing.assert_allclose(img, ref, rtol=1e-11, atol=1e-13)
                img1 = c.contrast_image(p, flow, 1)
                np.testing.assert_allclose(img1[0], ref[0], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("tiles", ["1", "2", "3", "7"])
def test_row_tiling_is_invisible(ebo_ab, orc, synth, tiles, monkeypatch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    ev, gt = synth.make_window(0, n_events=15000)
    monkeypatch.setenv("EBO_EVAL_TILES", tiles)
    with ctx_for(ebo, synth, 0) as c:
        c.set_window(ev)
        r, J = c.eval(gt * 0.7)
        ro, Jo, _, _ = orc.window_eval(ev, oparams(orc, c.params), gt * 0.7)
        assert_close(r[0], ro)
        assert_close(J[0], Jo, atol=1e-10)


@pytest.mark.parametrize("block", ["64", "128", "512"])
def test_block_size_is_invisible(ebo_ab, orc, synth, block, monkeypatch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    ev, gt = synth.make_window(0, n_events=15000)
    monkeypatch.setenv("EBO_EVAL_BLOCK", block)
    with ctx_for(ebo, synth, 0) as c:
        c.set_window(ev)
        r, J = c.eval(gt * 0.3)
        ro, Jo, _, _ = orc.window_eval(ev, oparams(orc, c.params), gt * 0.3)
        assert_close(r[0], ro)
        assert_close(J[0], Jo, atol=1e-10)


def test_out_of_window_penalty_branch(ebo, orc, synth):
    """Huge flows: every event leaves the 3x patch window (contrast_functor.h:143-149)."""
    ev, _ = synth.make_window(0, n_events=15000)
    with ctx_for(ebo, synth, 0) as c:
        c.set_window(ev)
        flows = np.tile(np.array([[3.0e3, -7.0e3]]), (c.P, 1))
        flows[5] = (1e9, 1e9)  # beyond int range: reference's int() is undefined, both skip
        r, J = c.eval(flows)
        ro, Jo, active, _ = orc.window_eval(ev, oparams(orc, c.params), flows)
        assert_close(r[0], ro, rtol=1e-15)
        assert_close(J[0], Jo, rtol=1e-15)
        a = np.nonzero(active)[0][0]
        assert r[0][a] == 1e3 * (1 + flows[a, 0] ** 2 + flows[a, 1] ** 2)


def test_central_difference_gradient_mode(ebo, orc, synth):
    ev, gt = synth.make_window(0, n_events=15000)
    h = 1e-6
    with ctx_for(ebo, synth, 0, grad=ebo.GRAD_CENTRAL, fd_step=h) as c:
        c.set_window(ev)
        flows = gt * 0.4
        r, J = c.eval(flows)
        prm = oparams(orc, c.params)
        ro, Jo, active, _ = orc.window_eval(ev, prm, flows)
        assert_close(r[0], ro)
        num = np.zeros_like(Jo)
        for k in range(2):
            d = np.zeros_like(flows)
            d[:, k] = h
            rp, _, _, _ = orc.window_eval(ev, prm, flows + d, want_jac=False)
            rm, _, _, _ = orc.window_eval(ev, prm, flows - d, want_jac=False)
            num[:, k] = (rp - rm) / (2 * h)
        # same formula on both sides: limited by cancellation in (r+ - r-), r ~ 1e3
        np.testing.assert_allclose(J[0], num, rtol=0, atol=2e-6)
        # and it approximates the analytic (Jet) Jacobian where no event changes bin
        close = np.abs(J[0] - Jo) < 1e-3
        assert close[active == 1].mean() > 0.9


def test_edge_cases_sparse_and_stray(ebo, orc):
    """Empty patches, patches at exactly min_events, events outside the sensor."""
    rng = np.random.RandomState(3)
    n = 400
    x = rng.randint(0, 20, n)
    y = rng.randint(0, 20, n)
    t = np.sort(rng.randint(0, 20000, n)) + 5000
    # patch (1,0): exactly 100 events (not > 100 => inactive); patch (2,0): 101 events
    x = np.concatenate([x, rng.randint(20, 40, 100), rng.randint(40, 60, 101), [-3, 250, 7, 300]])
    y = np.concatenate([y, rng.randint(0, 20, 100), rng.randint(0, 20, 101), [5, 5, -9, 200]])
    t = np.concatenate([t, np.sort(rng.randint(0, 20000, 100)) + 5000,
                        np.sort(rng.randint(0, 20000, 101)) + 5000, [6000, 7000, 8000, 9000]])
    order = np.argsort(t, kind="stable")
    ev = orc.make_events(x[order], y[order], t[order], np.where(rng.rand(len(t)) < 0.5, -1, 1))
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        c.set_window(ev)
        assert c.patch_info(0)[:2] == (400, True)
        assert c.patch_info(1)[:2] == (100, False)
        assert c.patch_info(2)[:2] == (101, True)
        assert c.patch_info(50)[:2] == (0, False)
        flows = rng.uniform(-0.5, 0.5, (c.P, 2))
        r, J = c.eval(flows)
        prm = oparams(orc, c.params)
        ro, Jo, active, _ = orc.window_eval(ev, prm, flows)
        assert active.sum() == 2
        assert_close(r[0], ro)
        assert_close(J[0], Jo, atol=1e-10)
        # count images with strays: bit exact
        assert np.array_equal(c.count_image(ebo.COUNT_INTEGRATED)[0], orc.integrate_events(ev, 240, 180))
        assert np.array_equal(c.count_image(ebo.COUNT_WARPED, flows)[0], orc.final_count_image(ev, prm, flows))


@pytest.mark.parametrize("config,n_events", [(0, 15000), (2, 50000), (4, 300000)])
def test_count_images_bit_exact(ebo, orc, synth, config, n_events):
    ev, gt = synth.make_window(config, n_events=n_events)
    with ctx_for(ebo, synth, config) as c:
        c.set_window(ev)
        prm = oparams(orc, c.params)
        w, h = c.params.image_w, c.params.image_h
        assert np.array_equal(c.count_image(ebo.COUNT_INTEGRATED)[0], orc.integrate_events(ev, w, h))
        rng = np.random.RandomState(11)
        for flows in (np.zeros((c.P, 2)), gt, rng.uniform(-3, 3, (c.P, 2))):
            img = c.count_image(ebo.COUNT_WARPED, flows)[0]
            ref = orc.final_count_image(ev, prm, flows)
            assert np.array_equal(img, ref)
            assert img.sum() <= len(ev)
        field = rng.uniform(-2, 2, (h, w, 2)).astype(np.float32)
        assert np.array_equal(c.count_image(ebo.COUNT_FIELD, field)[0],
                              orc.compensate_events_field(ev, w, h, field))
        # repeated calls start from a clean image
        assert np.array_equal(c.count_image(ebo.COUNT_INTEGRATED)[0], orc.integrate_events(ev, w, h))


def test_half_integer_rounding_on_device(ebo, orc):
    """round() half away from zero (feature_detector.cpp:446-453) on exact .5 positions."""
    # t_ref = 1000 us -> dt*scale = +1, 0, -1 ms; flows of +-0.5 and +-1.5 hit x.5 exactly
    xs, ys, ts = [], [], []
    for px in range(12):
        for k, t in enumerate((0, 1000, 2000)):
            xs.append(px * 20 + 5 + k)
            ys.append(7)
            ts.append(t)
    order = np.argsort(ts, kind="stable")
    ev = orc.make_events(np.array(xs)[order], np.array(ys)[order], np.array(ts)[order])
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        c.set_window(ev)
        flows = np.zeros((c.P, 2))
        flows[:12, 0] = [0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5]
        flows[:12, 1] = [-0.5, 0.5, -1.5, 1.5, 0.5, 0.5, -7.5, 7.5, 0.0, 0.0, 0.5, -0.5]
        img = c.count_image(ebo.COUNT_WARPED, flows)[0]
        assert np.array_equal(img, orc.final_count_image(ev, oparams(orc, c.params), flows))


def test_batch_of_windows_equals_one_by_one(ebo, orc, synth):
    ev, offsets, gt = synth.make_stream(0, 3, n_events=9000)
    with ctx_for(ebo, synth, 0, max_windows=3) as c:
        c.set_windows(ev, offsets)
        flows = gt * 0.5
        r, J = c.eval(flows)
        img = c.count_image(ebo.COUNT_WARPED, flows)
        prm = oparams(orc, c.params)
        for w in range(3):
            sub = ev[int(offsets[w]):int(offsets[w + 1])]
            ro, Jo, _, _ = orc.window_eval(sub, prm, flows[w])
            assert_close(r[w], ro)
            assert_close(J[w], Jo, atol=1e-10)
            assert np.array_equal(img[w], orc.final_count_image(sub, prm, flows[w]))
            assert c.window_info(w)[0] == orc.mid_timestamp(sub["t_us"][0], sub["t_us"][-1])


def _check_flows(f_gpu, f_ref, tol=1e-5):
    err = np.abs(np.asarray(f_gpu) - np.asarray(f_ref)).max()
    assert err <= tol, "max |flow_gpu - flow_oracle| = %.3e" % err


# The TV-free solve is chaotic in the reference's own formulation (residual
# 1000 - var makes every Gauss-Newton step ~1000/|J|): a 1-ulp change of
# compensateScale moves the CPU path's OWN answer by 0.2 px/ms after 50 iterations
# (tests/test_oracle_golden.py::test_tv_free_solve_is_chaotic_on_the_cpu_alone).  The
# 1e-5 bar is therefore checked (a) on the reference's configuration (TV on, full 50
# iterations) and (b) on TV-free solves up to the iteration count where the CPU path's
# own 1-ulp sensitivity is still below the bar; past that, on invariants.
@pytest.mark.parametrize("config,n_events", [(0, 15000), (2, 50000)])
@pytest.mark.parametrize("iters", [1, 4, 8, 12, 16, 20])
def test_solve_independent_on_device(ebo, orc, synth, config, n_events, iters):
    """Per-patch LM entirely on the device vs the oracle's per-patch LM."""
    ev, _ = synth.make_window(config, n_events=n_events)
    with ctx_for(ebo, synth, config, tv_weight=0.0) as c:
        c.set_window(ev)
        flows, summ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=iters)
        prm = oparams(orc, c.params)
        fo, _, so = orc.compensate_events_contrast(
            ev, prm, orc.default_solver(mode=1, max_num_iterations=iters), want_image=False)
        _check_flows(flows[0], fo)
        # same decisions: same number of cost-only and Jacobian evaluations
        assert summ[0].num_evals_jac == so.num_evals_jac
        assert summ[0].num_evals_cost == so.num_evals_cost
        assert summ[0].iterations == so.iterations


def test_solve_independent_full_run_invariants(ebo, orc, synth):
    """Full 50 iterations, TV-free: both sides are past the chaos horizon, so compare
    what is well defined: the objective at the returned flows (recomputed by the
    oracle) never exceeds the objective at the start, and the device's solution is as
    good as the oracle's on aggregate."""
    ev, _ = synth.make_window(2, n_events=50000)
    with ctx_for(ebo, synth, 2, tv_weight=0.0) as c:
        c.set_window(ev)
        flows, summ = c.solve(mode=ebo.SOLVE_INDEPENDENT)
        prm = oparams(orc, c.params)
        fo, _, so = orc.compensate_events_contrast(ev, prm, orc.default_solver(mode=1), want_image=False)
        assert np.isfinite(flows).all()
        r0, _, active, _ = orc.window_eval(ev, prm, np.zeros((c.P, 2)), want_jac=False)
        rg, _, _, _ = orc.window_eval(ev, prm, flows[0], want_jac=False)
        ro, _, _, _ = orc.window_eval(ev, prm, fo, want_jac=False)
        a = active == 1
        assert np.all(rg[a] <= r0[a] + 1e-9)  # LM returns the lowest-cost point visited
        # aggregate quality comparable to the oracle's own (variance gain = r0 - r); the
        # two chaotic runs differ by ~10 % in either direction
        gain_g, gain_o = (r0[a] - rg[a]).sum(), (r0[a] - ro[a]).sum()
        assert gain_g > 0 and gain_o > 0
        assert 0.7 <= gain_g / gain_o <= 1.0 / 0.7
        # most patches still coincide to 1e-5 even after 50 iterations
        d = np.abs(flows[0] - fo).max(axis=1)
        assert (d <= 1e-5).mean() >= 0.5
        assert summ[0].iterations <= 50 and summ[0].termination in (0, 1)


def test_solve_global_with_tv_matches_reference_problem(ebo, orc, synth):
    """The reference's problem: all patches + TV terms in one LM (host) with batched
    device evaluations (feature_detector.cpp:316-414)."""
    ev, _ = synth.make_window(0, n_events=15000)
    with ctx_for(ebo, synth, 0) as c:  # tv_weight 1e3, huber 10 (defaults)
        c.set_window(ev)
        flows, summ = c.solve(mode=ebo.SOLVE_GLOBAL)
        prm = oparams(orc, c.params)
        fo, _, so = orc.compensate_events_contrast(ev, prm, orc.default_solver(mode=0), want_image=False)
        _check_flows(flows[0], fo)
        assert summ[0].iterations == so.iterations
        assert summ[0].termination == so.termination
        assert summ[0].final_cost == pytest.approx(so.final_cost, rel=1e-9)


@pytest.mark.parametrize("iters", [6, 12, 18])
def test_solve_global_without_tv(ebo, orc, synth, iters):
    """One global trust region over all patches, no TV (host LM + device evaluations)."""
    ev, _ = synth.make_window(0, n_events=12000)
    with ctx_for(ebo, synth, 0, tv_weight=0.0) as c:
        c.set_window(ev)
        flows, summ = c.solve(mode=ebo.SOLVE_GLOBAL, max_num_iterations=iters)
        fo, _, so = orc.compensate_events_contrast(
            ev, oparams(orc, c.params), orc.default_solver(mode=0, max_num_iterations=iters),
            want_image=False)
        _check_flows(flows[0], fo)
        assert summ[0].iterations == so.iterations
        assert summ[0].num_evals_jac == so.num_evals_jac


def test_compensate_events_contrast_one_call(ebo, orc, synth):
    """R2 end to end: flows + final count image."""
    ev, _ = synth.make_window(0, n_events=15000)
    with ctx_for(ebo, synth, 0) as c:
        flows, img, s = c.compensate_events_contrast(ev)
        prm = oparams(orc, c.params)
        fo, io, so = orc.compensate_events_contrast(ev, prm, orc.default_solver())
        _check_flows(flows, fo)
        # count image is bit exact for the SAME flows
        assert np.array_equal(img, orc.final_count_image(ev, prm, flows))
        # and with each side's own flows at most a few events land one pixel apart
        assert np.abs(img - io).sum() <= 0.002 * len(ev)


def test_patch_integrate_reference_known_answer(ebo, orc):
    """patch_test.cpp:35-60 through the HIP path."""
    xs = [7 + i // 7 for i in range(30)]
    ys = [7 + i % 7 for i in range(30)]
    sg = [1 if i % 2 == 0 else -1 for i in range(30)]
    ev = orc.make_events(xs, ys, list(range(30)), sg)[::-1].copy()
    with ebo.Context() as c:
        imgs, cur, last = c.patch_integrate(ev, [0, 30], [(7.0, 7.0, 7.0, 7.0)])
        nabla = imgs[0]
        for i in range(30):
            assert nabla[i % 7, i // 7] == (1.0 if i % 2 == 0 else -1.0)
        assert cur[0] == 14 and last[0] == 0


def test_patch_integrate_batched(ebo, orc):
    rng = np.random.RandomState(5)
    evs, offs, rects, trajs, mids = [], [0], [], [], []
    for k in range(40):
        n = rng.randint(1, 300)
        cx, cy = rng.uniform(20, 200), rng.uniform(20, 150)
        ext = 12
        rect = (np.floor(cx) - ext + (0.5 if k % 3 == 0 else 0.0), np.floor(cy) - ext, 2 * ext + 1, 2 * ext + 1)
        x = rng.randint(int(cx) - 16, int(cx) + 17, n)
        y = rng.randint(int(cy) - 16, int(cy) + 17, n)
        t = np.sort(rng.randint(0, 30000, n))[::-1] + 1000  # deque order: newest first
        e = orc.make_events(x, y, t, np.where(rng.rand(n) < 0.5, -1, 1))
        evs.append(e)
        offs.append(offs[-1] + n)
        rects.append(rect)
        t_last = int(t[0]) - rng.randint(0, 2000)
        t_pre = t_last - rng.randint(1000, 20000)
        trajs.append((cx - rng.uniform(-3, 3), cy - rng.uniform(-3, 3), t_pre, cx, cy, t_last))
        mids.append(orc.mid_timestamp(t[0], t[-1]) if k % 5 else t_last + 10 ** 7)
    ev = np.concatenate(evs)
    with ebo.Context() as c:
        imgs, cur, last = c.patch_integrate(ev, offs, rects)
        for k in range(40):
            ref, rc, rl = orc.patch_integrate(evs[k], rects[k])
            assert np.array_equal(imgs[k], ref)
            assert cur[k] == rc and last[k] == rl
        imgs, upd = c.patch_integrate_mc(ev, offs, rects, trajs, mids)
        n_upd = 0
        for k in range(40):
            tr = trajs[k]
            ref, ru = orc.patch_integrate_mc(evs[k], rects[k], tr[0:3], tr[3:6], mids[k])
            assert bool(upd[k]) == ru
            if ru:
                n_upd += 1
                assert np.array_equal(imgs[k], ref)
            else:
                assert not imgs[k].any()
        assert 0 < n_upd < 40


def test_error_behaviour(ebo, synth):
    ev, _ = synth.make_window(0, n_events=2000)
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        with pytest.raises(ebo.EboError) as ei:
            c.n_windows = 1
            c.eval(np.zeros((c.P, 2)))
        assert ei.value.code == ebo.ERR_STATE
        bad = ev.copy()
        bad["x"][3] = 40000
        with pytest.raises(ebo.EboError) as ei:
            c.set_window(bad)
        assert ei.value.code == ebo.ERR_RANGE
        far = ev.copy()
        far["t_us"][-1] += 1 << 33
        with pytest.raises(ebo.EboError) as ei:
            c.set_window(far)
        assert ei.value.code == ebo.ERR_RANGE
        with pytest.raises(ebo.EboError) as ei:
            c.set_windows(ev, [0, 1000, 2000])  # capacity is one window
        assert ei.value.code == ebo.ERR_ARG
        c.set_window(ev)  # still usable after errors
        r, _ = c.eval(np.zeros((c.P, 2)))
        assert r.shape == (1, c.P)


def test_evaluation_rounds_with_and_without_zero_copy(ebo_ab, synth, monkeypatch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Host-driven solves hand flows and results through pinned memory the kernels access directly
    (small rounds) or through explicit async copies (large ones, EBO_ZERO_COPY_MAX): same bits (variance
    loss: its evaluation is bit-reproducible; the edge loss's f64 LDS atomics are not)."""
    ev, offsets, _ = synth.make_stream(0, 3, n_events=6000)
    out = []
    for limit in ("4096", "0"):
        monkeypatch.setenv("EBO_ZERO_COPY_MAX", limit)
        with ebo.Context(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_VARIANCE,
                         max_events=len(ev), max_windows=3) as c:
            c.set_windows(ev, offsets)
            opts = ebo.default_solver()
            opts.max_num_iterations = 6
            flows, summ = c.solve(opts)
            r, J = c.eval(flows)
            out.append((flows.copy(), r.copy(), J.copy(), [s.final_cost for s in summ]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]


@pytest.mark.parametrize("loss", ["variance", "edge"])
@pytest.mark.parametrize("n", [1, 5, 19])
def test_speculative_jacobians_in_lock_step_keep_the_bits(ebo_ab, synth, monkeypatch, loss, n):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Once few windows are running, a window that asks for the cost at a candidate is evaluated with its Jacobian
    as well, and the solver's next request -- value and Jacobian at that very point, if it accepts the step -- is
    answered from that evaluation without another round.  The solver sees the same numbers in the same order: flows,
    iteration and evaluation counts, final costs equal the unspeculative solve's bit for bit (one window: the plain
    driver; 5 and 19: two groups in flight)."""
    ev, offsets, _ = synth.make_stream(0, n)
    kw = dict(image_w=240, image_h=180, patch_w=20, patch_h=20,
              loss=ebo.LOSS_VARIANCE if loss == "variance" else ebo.LOSS_EDGE, max_events=len(ev), max_windows=n)
    out = []
    for spec in ("0", None):
        if spec is None:
            monkeypatch.delenv("EBO_SOLVE_SPECULATE", raising=False)
        else:
            monkeypatch.setenv("EBO_SOLVE_SPECULATE", spec)
        with ebo.Context(**kw) as c:
            c.set_windows(ev, offsets)
            opts = ebo.default_solver()
            opts.max_num_iterations = 15
            flows, summ = c.solve(opts)
            out.append((flows.copy(), [(s.iterations, s.final_cost, s.termination, s.num_evals_cost, s.num_evals_jac) for s in summ]))
    monkeypatch.delenv("EBO_SOLVE_SPECULATE", raising=False)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert max(s[4] for s in out[0][1]) > 1  # accepted steps happened: Jacobian requests beyond the first


@pytest.mark.parametrize("loss", ["variance", "edge"])
@pytest.mark.parametrize("config", [0, 2])
def test_device_solve_reusing_the_image_of_an_accepted_step_keeps_the_bits(ebo_ab, synth, monkeypatch, loss, config):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """The device-resident per-patch solve evaluates the cost at a candidate and, on acceptance, value and Jacobian
    at the same point; the second evaluation starts from what the first left in LDS (variance loss: the image, gather
    pass only; edge loss: image, eigenvalues and directions, from the window maxima on).  Same operations on the
    same operands: flows, iteration and evaluation counts, terminations equal the full evaluations' bit for bit."""
    n = 6 if config == 0 else 3
    ev, offsets, _ = synth.make_stream(config, n)
    cfg = synth.CONFIGS[config]
    kw = dict(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
              loss=ebo.LOSS_VARIANCE if loss == "variance" else ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=n)
    out = []
    for no_reuse in (False, True):
        if no_reuse:
            monkeypatch.setenv("EBO_SOLVE_NO_REUSE", "1")
        else:
            monkeypatch.delenv("EBO_SOLVE_NO_REUSE", raising=False)
        with ebo.Context(**kw) as c:
            c.set_windows(ev, offsets)
            opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT)
            opts.max_num_iterations = 12
            flows, summ = c.solve(opts)
            out.append((flows.copy(), [(s.iterations, s.termination, s.num_evals_cost, s.num_evals_jac) for s in summ]))
    monkeypatch.delenv("EBO_SOLVE_NO_REUSE", raising=False)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert max(s[3] for s in out[0][1]) > 1  # Jacobian evaluations beyond the first: accepted steps, i.e. reuse happened


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["variance", "edge"])
@pytest.mark.parametrize("n", [6, 41])
def test_thinned_out_rounds_as_window_lists_keep_the_bits(ebo_ab, synth, monkeypatch, loss, n):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Late in a lock-step solve few windows of a batch are still running; a round then launches workgroups for
    those windows' units only (a list in the kernel arguments, flows and results through pinned memory) instead
    of every unit with a mode table and copies.  Same evaluations of the same points: flows, iteration counts and
    final costs equal the list-free path bit for bit, in the plain (6 windows) and the pipelined (41) driver."""
    ev, offsets, _ = synth.make_stream(0, n)
    kw = dict(image_w=240, image_h=180, patch_w=20, patch_h=20,
              loss=ebo.LOSS_VARIANCE if loss == "variance" else ebo.LOSS_EDGE, max_events=len(ev), max_windows=n)
    out = []
    for no_lists in (False, True):
        if no_lists:
            monkeypatch.setenv("EBO_SOLVE_NO_COMPACT", "1")
        else:
            monkeypatch.delenv("EBO_SOLVE_NO_COMPACT", raising=False)
        with ebo.Context(**kw) as c:
            c.set_windows(ev, offsets)
            opts = ebo.default_solver()
            opts.max_num_iterations = 15
            flows, summ = c.solve(opts)
            out.append((flows.copy(), [(s.iterations, s.final_cost, s.termination, s.num_evals_cost, s.num_evals_jac) for s in summ]))
    monkeypatch.delenv("EBO_SOLVE_NO_COMPACT", raising=False)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert len({s[0] for s in out[0][1]}) > 1  # the windows do not all stop together: some rounds were thinned out


@pytest.mark.parametrize("loss", ["variance", "edge"])
def test_central_difference_rounds_never_take_the_window_list(ebo_ab, synth, monkeypatch, loss):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """A central-difference Jacobian round launches five flow sets and a combine kernel over EVERY unit; those
    launches know no window list.  Such rounds must take the full path (which copies back only the half's own
    slots): in the pipelined driver (41 windows, two halves in flight) the solve equals, bit for bit, the
    list-free and the unpipelined one."""
    n = 41
    ev, offsets, _ = synth.make_stream(0, n)
    kw = dict(image_w=240, image_h=180, patch_w=20, patch_h=20, grad=ebo.GRAD_CENTRAL,
              loss=ebo.LOSS_VARIANCE if loss == "variance" else ebo.LOSS_EDGE, max_events=len(ev), max_windows=n)
    out = []
    for env in (None, "EBO_SOLVE_NO_COMPACT", "EBO_SOLVE_NO_PIPELINE"):
        for k in ("EBO_SOLVE_NO_COMPACT", "EBO_SOLVE_NO_PIPELINE"):
            monkeypatch.delenv(k, raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        with ebo.Context(**kw) as c:
            c.set_windows(ev, offsets)
            opts = ebo.default_solver()
            opts.max_num_iterations = 30
            flows, summ = c.solve(opts)
            out.append((flows.copy(), [(s.iterations, s.final_cost, s.termination, s.num_evals_cost, s.num_evals_jac) for s in summ]))
    for k in ("EBO_SOLVE_NO_COMPACT", "EBO_SOLVE_NO_PIPELINE"):
        monkeypatch.delenv(k, raising=False)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[2][0]) and out[0][1] == out[2][1]
    assert len({s[0] for s in out[0][1]}) > 1  # windows stop at different rounds: thinned-out rounds happened


def test_pipelined_lock_step_solve_equals_the_plain_one(ebo_ab, synth, monkeypatch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """With four or more windows the TV-coupled host LM runs two halves in flight (one half's LM steps
    on the host while the device evaluates the other): per window the same requests in the same
    order, so the same bits as the one-round-at-a-time loop; and the same answer as a window alone."""
    n = 21
    ev, offsets, _ = synth.make_stream(0, n, n_events=5000)
    kw = dict(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_VARIANCE, max_events=len(ev), max_windows=n)
    out = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("EBO_SOLVE_NO_PIPELINE", "1")
        else:
            monkeypatch.delenv("EBO_SOLVE_NO_PIPELINE", raising=False)
        with ebo.Context(**kw) as c:
            c.set_windows(ev, offsets)
            opts = ebo.default_solver()
            opts.max_num_iterations = 12
            flows, summ = c.solve(opts)
            out.append((flows.copy(), [(s.iterations, s.final_cost, s.termination) for s in summ]))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    monkeypatch.delenv("EBO_SOLVE_NO_PIPELINE", raising=False)
    for w in (0, 10, 20):
        sub = ev[int(offsets[w]):int(offsets[w + 1])]
        with ebo.Context(**dict(kw, max_events=len(sub), max_windows=1)) as c:
            c.set_window(sub)
            opts = ebo.default_solver()
            opts.max_num_iterations = 12
            alone, s1 = c.solve(opts)
        np.testing.assert_allclose(alone[0], out[0][0][w], rtol=0, atol=1e-7)
        assert s1[0].iterations == out[0][1][w][0]


def test_text_to_device_through_compact_records(ebo, synth, tmp_path):
    """Round 5: ebo_read_events_txt8 parses an events.txt straight into the 8-byte records ebo_set_windows8 takes (base time =
    the first event's): the windows loaded that way -- unit tables, objective, Jacobian, count images -- are those of the
    24-byte path (ebo_read_events_txt + ebo_set_windows) bit for bit."""
    ev, offsets, gt = synth.make_stream(0, 3)
    p = tmp_path / "events.txt"
    with open(p, "w") as f:
        f.write("".join("%d.%06d %d %d %d\n" % (t // 1000000, t % 1000000, x, y, 1 if s > 0 else 0)
                        for t, x, y, s in zip(ev["t_us"].tolist(), ev["x"].tolist(), ev["y"].tolist(), ev["sign"].tolist())))
    ev24 = ebo.read_events_txt(str(p), cap=len(ev) + 8)
    # seconds -> double -> x 1e6 -> truncation, as the reference's reader does: a microsecond may be lost on the way in
    assert np.abs(ev24["t_us"] - ev["t_us"]).max() <= 1 and np.array_equal(ev24["x"], ev["x"]) and np.array_equal(ev24["sign"], ev["sign"])
    ev8, base, off = ebo.read_events_txt8(str(p), len(ev) + 8, threads=3, offset=0)
    assert len(ev8) == len(ev) and base == int(ev24["t_us"][0]) and off == os.path.getsize(p)
    assert np.array_equal(ev8, ebo.pack_events8(ev24, base))
    flows = gt * 0.4
    with ebo.Context(loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev), max_windows=3) as a, \
            ebo.Context(loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev), max_windows=3) as b:
        a.set_windows(ev24, offsets)
        b.set_windows8(ev8, [base] * 3, offsets)
        ra, Ja = a.eval(flows)
        rb, Jb = b.eval(flows)
        assert np.array_equal(ra, rb) and np.array_equal(Ja, Jb)
        for mode in (ebo.COUNT_INTEGRATED, ebo.COUNT_WARPED):
            assert np.array_equal(a.count_image(mode, flows if mode == ebo.COUNT_WARPED else None),
                                  b.count_image(mode, flows if mode == ebo.COUNT_WARPED else None))
        for q in range(a.P):
            assert a.patch_info(q, 1) == b.patch_info(q, 1)
