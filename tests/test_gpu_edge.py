"""GPU parity tests of the edge (structure-tensor) loss — the loss the reference's
contrastFunctor::operator() actually calls (contrast_functor.h:33-34, :152-277):
value, Jacobian (what Jet<double,2> propagates), penalty branch, fallback storage,
central differences, and the solves on the reference's own configuration."""
import json
import os

import numpy as np
import pytest

from test_oracle_golden import GOLDEN, make_probe_input

pytestmark = pytest.mark.gpu


def oparams(orc, p):
    return orc.default_params(
        image_w=p.image_w, image_h=p.image_h, patch_w=p.patch_w, patch_h=p.patch_h,
        tv_weight=p.tv_weight, tv_huber=p.tv_huber, scale=p.scale, min_events=p.min_events,
        loss=p.loss)


def ctx_for(ebo, synth, config, **kw):
    cfg = synth.CONFIGS[config]
    args = dict(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                patch_h=cfg["patch"][1], loss=ebo.LOSS_EDGE)
    args.update(kw)
    return ebo.Context(**args)


def check_rj(r, J, ro, Jo):
    np.testing.assert_allclose(r, ro, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(J, Jo, rtol=1e-8, atol=1e-12)


def test_probe_digits_edge_loss_through_the_hip_path(ebo, orc):
    """SURVEY §8(c) digits of the reference's edge loss, evaluated by the HIP kernels."""
    gold = json.load(open(os.path.join(GOLDEN, "survey_probe_contrast.json")))
    ev = make_probe_input(orc)
    with ebo.Context(loss=ebo.LOSS_EDGE) as c:  # the reference's default loss
        c.set_patches(ev, [0, len(ev)], [gold["input"]["patch_rect"]])
        for case in gold["cases"]:
            r, J = c.eval([case["m"]])
            assert r[0, 0] == pytest.approx(case["edge_r"], rel=1e-12)
            np.testing.assert_allclose(J[0, 0], case["edge_J"], rtol=1e-8, atol=1e-14)
            r1, _ = c.eval([case["m"]], want_jac=False)
            assert r1[0, 0] == pytest.approx(case["edge_r"], rel=1e-12)


@pytest.mark.parametrize("config,n_events", [(0, 15000), (2, 30000)])
def test_edge_eval_value_and_jacobian(ebo, orc, synth, config, n_events):
    ev, gt = synth.make_window(config, n_events=n_events)
    with ctx_for(ebo, synth, config) as c:
        c.set_window(ev)
        rng = np.random.RandomState(5 + config)
        prm = oparams(orc, c.params)
        for flows in (np.zeros((c.P, 2)), gt * 0.5, rng.uniform(-0.6, 0.6, (c.P, 2))):
            r, J = c.eval(flows)
            ro, Jo, active, _ = orc.window_eval(ev, prm, flows)
            check_rj(r[0], J[0], ro, Jo)
            r1, _ = c.eval(flows, want_jac=False)
            np.testing.assert_allclose(r1[0], ro, rtol=1e-9)
            assert np.all(r[0][active == 0] == 0.0)


@pytest.mark.parametrize("forms,layout", [("0", "0"), ("0", "1"), ("1", "0"), ("7", "0"), ("7", "1")])
def test_edge_tensor_filter_forms_and_layouts(ebo_ab, orc, synth, monkeypatch, forms, layout):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Every selectable form of the structure-tensor filter (direct 49-tap, band buffers, register
    runs) on both LDS layouts gives the oracle's value and Jacobian (random flows: no exact ties)."""
    monkeypatch.setenv("EBO_EDGE_SEPARABLE", forms)
    monkeypatch.setenv("EBO_EDGE_LAYOUT", layout)
    ev, gt = synth.make_window(0, n_events=15000)
    with ctx_for(ebo, synth, 0) as c:
        c.set_window(ev)
        prm = oparams(orc, c.params)
        flows = np.random.RandomState(11).uniform(-0.8, 0.8, (c.P, 2))
        r, J = c.eval(flows)
        ro, Jo, _, _ = orc.window_eval(ev, prm, flows)
        check_rj(r[0], J[0], ro, Jo)
        r1, _ = c.eval(flows, want_jac=False)
        np.testing.assert_allclose(r1[0], ro, rtol=1e-9)


def test_edge_penalty_branch_and_sparse_images(ebo, orc, synth):
    """mean(image) <= 1e-4 => 1e3 (1 + |m|^2) (contrast_functor.h:159-165): huge flows, and
    flows that leave only a handful of events inside the window."""
    ev, gt = synth.make_window(0, n_events=15000)
    with ctx_for(ebo, synth, 0) as c:
        c.set_window(ev)
        prm = oparams(orc, c.params)
        taken = []
        for flows in (np.tile([[2.0e3, -3.0e3]], (c.P, 1)), np.tile([[1.6, -1.9]], (c.P, 1)),
                      np.tile([[3.1, 2.7]], (c.P, 1))):
            r, J = c.eval(flows)
            ro, Jo, active, _ = orc.window_eval(ev, prm, flows)
            check_rj(r[0], J[0], ro, Jo)
            taken.append(int((ro[active == 1] > 1e3).sum()))
        # huge flow: (nearly) every patch pays the penalty — a patch keeps its image only
        # if enough of its events sit at dt ~ 0; moderate flows: none does
        assert taken[0] >= 0.9 * (active == 1).sum()
        assert taken[1] == 0 and taken[2] == 0


def test_edge_global_memory_fallback(ebo_ab, orc, synth, monkeypatch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Boxes that do not fit LDS use the per-unit global slice: same numbers."""
    ev, gt = synth.make_window(2, n_events=30000)
    monkeypatch.setenv("EBO_EDGE_LDS_KB", "24")  # far too small for a 30x22 patch's box
    with ctx_for(ebo, synth, 2) as c:
        c.set_window(ev)
        flows = gt * 0.5
        r, J = c.eval(flows)
        ro, Jo, _, _ = orc.window_eval(ev, oparams(orc, c.params), flows)
        check_rj(r[0], J[0], ro, Jo)


def test_edge_central_difference_mode(ebo, orc, synth):
    ev, gt = synth.make_window(0, n_events=15000)
    h = 1e-6
    with ctx_for(ebo, synth, 0, grad=ebo.GRAD_CENTRAL, fd_step=h) as c:
        c.set_window(ev)
        flows = gt * 0.4
        r, J = c.eval(flows)
        prm = oparams(orc, c.params)
        ro, _, _, _ = orc.window_eval(ev, prm, flows, want_jac=False)
        np.testing.assert_allclose(r[0], ro, rtol=1e-9)
        num = np.zeros((c.P, 2))
        for k in range(2):
            d = np.zeros_like(flows)
            d[:, k] = h
            rp, _, _, _ = orc.window_eval(ev, prm, flows + d, want_jac=False)
            rm, _, _, _ = orc.window_eval(ev, prm, flows - d, want_jac=False)
            num[:, k] = (rp - rm) / (2 * h)
        np.testing.assert_allclose(J[0], num, rtol=0, atol=2e-6)


def test_reference_configuration_end_to_end(ebo, orc, synth):
    """THE reference call: compensateEventsContrast with every default (edge loss, Jet
    gradient, TV 1e3 / Huber 10, one global problem, 50 iterations) — flows within 1e-5,
    same iteration count, final count image bit-exact for the returned flows."""
    ev, _ = synth.make_window(0, n_events=15000)
    with ebo.Context() as c:  # all defaults == DetectorParams defaults
        assert c.params.loss == ebo.LOSS_EDGE
        flows, img, s = c.compensate_events_contrast(ev)
        prm = orc.default_params()
        fo, io, so = orc.compensate_events_contrast(ev, prm, orc.default_solver())
        err = np.abs(flows - fo).max()
        assert err <= 1e-5, err
        assert s.iterations == so.iterations and s.termination == so.termination
        assert s.final_cost == pytest.approx(so.final_cost, rel=1e-9)
        assert np.array_equal(img, orc.final_count_image(ev, prm, flows))
        assert np.abs(img - io).sum() <= 0.002 * len(ev)


@pytest.mark.parametrize("how", ["device", "lockstep"])
@pytest.mark.parametrize("iters", [4, 8, 10])
def test_edge_independent_solve_lockstep(ebo_ab, orc, synth, monkeypatch, iters, how):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Per-patch problems (TV off) with the edge loss, the reference's own objective: the whole LM
    in one launch (k_solve_edge, the default) and, for A/B, host LMs in lock step over batched
    device evaluations; capped below the chaos horizon (DESIGN.md section 2), which is shorter
    for the edge loss: its derivative jumps when a window's argmax moves."""
    monkeypatch.setenv("EBO_SOLVE_EDGE", how)
    ev, _ = synth.make_window(0, n_events=15000)
    with ctx_for(ebo, synth, 0, tv_weight=0.0) as c:
        c.set_window(ev)
        flows, summ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=iters)
        fo, _, so = orc.compensate_events_contrast(
            ev, oparams(orc, c.params), orc.default_solver(mode=1, max_num_iterations=iters),
            want_image=False)
        err = np.abs(flows[0] - fo).max()
        assert err <= 1e-5, err
        assert summ[0].iterations == so.iterations
        assert summ[0].num_evals_jac == so.num_evals_jac
        assert summ[0].num_evals_cost == so.num_evals_cost


@pytest.mark.parametrize("config,windows", [(0, 8), (2, 2), (3, 1)])
def test_edge_device_solve_equals_the_lockstep_solve(ebo_ab, synth, monkeypatch, config, windows):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """ebo_solve_device with the edge loss (no EBO_ERR_UNSUPPORTED any more): the same flows and
    per-patch statistics as the host-driven lock-step solve of the same windows, run to run
    identical bits, also where boxes spill into the global-memory slices (C3's corner patches)."""
    import ctypes as C
    cfg = synth.CONFIGS[config]
    ev, offsets, _ = synth.make_stream(config, windows)
    hip = C.CDLL("libamdhip64.so")
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                     patch_h=cfg["patch"][1], loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev),
                     max_windows=windows) as c:
        c.set_windows(ev, offsets)
        n = windows * c.P
        d_sol, d_st = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(d_sol), C.c_size_t(n * 16)) == 0
        assert hip.hipMalloc(C.byref(d_st), C.c_size_t(n * 16)) == 0
        opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=8)
        got = []
        for _ in range(2):
            c.solve_device(opts, d_sol.value, d_st.value)
            c.synchronize()
            sol, st = np.zeros((windows, c.P, 2)), np.zeros((windows, c.P, 4), dtype=np.int32)
            assert hip.hipMemcpy(sol.ctypes.data_as(C.c_void_p), d_sol, C.c_size_t(n * 16), 2) == 0
            assert hip.hipMemcpy(st.ctypes.data_as(C.c_void_p), d_st, C.c_size_t(n * 16), 2) == 0
            got.append((sol, st))
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
        monkeypatch.setenv("EBO_SOLVE_EDGE", "lockstep")
        ref, summ = c.solve(opts)
        sol, st = got[0]
        np.testing.assert_allclose(sol, ref, rtol=0, atol=1e-9)
        for w in range(windows):
            assert st[w, :, 0].max() == summ[w].iterations
            assert st[w, :, 1].sum() == summ[w].num_evals_cost and st[w, :, 2].sum() == summ[w].num_evals_jac
        hip.hipFree(d_sol)
        hip.hipFree(d_st)


@pytest.mark.parametrize("config,windows", [(0, 16), (2, 4), (3, 2)])
def test_edge_jacobian_is_bit_reproducible(ebo, synth, config, windows):
    """The adjoint image of the reverse pass is accumulated in exact fixed point on a per-unit grid
    (integer LDS atomics commute), so value AND Jacobian are identical bits run to run, whatever
    order the workgroup's atomics retire in."""
    cfg = synth.CONFIGS[config]
    ev, offsets, gt = synth.make_stream(config, windows)
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                     patch_h=cfg["patch"][1], loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev),
                     max_windows=windows) as c:
        c.set_windows(ev, offsets)
        for scale in (0.0, 0.5, 1.0):
            r0, J0 = c.eval(gt * scale)
            assert np.isfinite(J0).all() and np.abs(J0).max() > 0
            for _ in range(4):
                r, J = c.eval(gt * scale)
                assert np.array_equal(r, r0) and np.array_equal(J, J0)


@pytest.mark.parametrize("config,windows", [(0, 8), (2, 2), (3, 2)])
def test_edge_direction_table_equals_rederived_tensor_sums(ebo, orc, synth, monkeypatch, config, windows):
    """A Jacobian evaluation's reverse pass reads each argmax pixel's eigenvector direction from the
    table the eigenvalue pass wrote (global memory, EBO_EDGE_CS_MB) instead of re-deriving the
    tensor sums: the same Jacobian to rounding, the same value bit for bit, and both within the
    oracle's bar (zero flow included: units whose d = 0 flag their Jacobian NaN on both paths)."""
    cfg = synth.CONFIGS[config]
    ev, offsets, gt = synth.make_stream(config, windows)
    got = {}
    for mb in ("4096", "0"):
        monkeypatch.setenv("EBO_EDGE_CS_MB", mb)
        with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                         patch_h=cfg["patch"][1], loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev),
                         max_windows=windows) as c:
            c.set_windows(ev, offsets)
            got[mb] = [c.eval(gt * s) for s in (0.0, 0.5, 1.3)]
            if mb == "4096":
                prm = oparams(orc, c.params)
                sub = ev[int(offsets[0]):int(offsets[1])]
                ro, Jo, _, _ = orc.window_eval(sub, prm, gt[0] * 0.5)
                check_rj(got[mb][1][0][0], got[mb][1][1][0], ro, Jo)
    for (ra, Ja), (rb, Jb) in zip(got["4096"], got["0"]):
        assert np.array_equal(ra, rb)
        assert np.array_equal(np.isnan(Ja), np.isnan(Jb))
        ok = ~np.isnan(Ja)
        scale = np.abs(Jb[ok]).max()
        assert np.abs(Ja[ok] - Jb[ok]).max() <= 1e-11 * scale


@pytest.mark.parametrize("config,windows,kb", [(0, 6, "52"), (0, 6, "40"), (3, 2, "52")])
def test_edge_compact_layout_and_deferred_units_keep_the_bits(ebo_ab, orc, synth, monkeypatch, config, windows, kb):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """Round 5: batches larger than the chip holds run two launches -- every unit on the compact LDS layout (16.5 B per
    pixel: three workgroups per CU), units whose box does not fit deferred to a second launch on the 20 B layout.  Values
    and Jacobians equal the single-launch path BIT FOR BIT at small, mid-solve and wild flows (all units compact / a mix /
    mostly deferred, some of them on the global slice), value-only launches included, and the oracle to the usual bars."""
    cfg = synth.CONFIGS[config]
    ev, offsets, gt = synth.make_stream(config, windows)
    rng = np.random.default_rng(5)
    with ctx_for(ebo, synth, config, tv_weight=0.0, max_events=len(ev), max_windows=windows) as c:
        c.set_windows(ev, offsets)
        for scale in (0.0, 0.5, 1.0, 3.0):
            flows = gt * scale + (0.0 if scale == 0.0 else 0.01 * rng.standard_normal(gt.shape))
            monkeypatch.setenv("EBO_EDGE_COMPACT", "0")
            r0, J0 = c.eval(flows)
            v0, _ = c.eval(flows, want_jac=False)
            monkeypatch.setenv("EBO_EDGE_COMPACT", "1")
            monkeypatch.setenv("EBO_EDGE_COMPACT_KB", kb)
            r1, J1 = c.eval(flows)
            v1, _ = c.eval(flows, want_jac=False)
            r2, J2 = c.eval(flows)  # and run to run
            # ... and with the bounding boxes and the second launch's list made inside the compact launch (as until
            # k_edge_classify made them up front)
            monkeypatch.setenv("EBO_EDGE_CLASSIFY", "0")
            r3, J3 = c.eval(flows)
            v3, _ = c.eval(flows, want_jac=False)
            monkeypatch.delenv("EBO_EDGE_CLASSIFY")
            monkeypatch.delenv("EBO_EDGE_COMPACT_KB")
            assert np.array_equal(r0, r1) and np.array_equal(J0, J1, equal_nan=True), scale
            assert np.array_equal(v0, v1) and np.array_equal(r1, r2) and np.array_equal(J1, J2, equal_nan=True), scale
            assert np.array_equal(r3, r1) and np.array_equal(J3, J1, equal_nan=True) and np.array_equal(v3, v1), scale
            if scale in (0.5, 1.0):
                ro, Jo, _, _ = orc.window_eval(ev[offsets[0]:offsets[1]], oparams(orc, c.params), flows[0])
                check_rj(r1[0], J1[0], ro, Jo)


def test_edge_large_batch_takes_the_compact_path_by_itself(ebo, orc, synth):
    """The shipped library, no switch: 24 reference-default windows = 2616 units > eight per CU, so the evaluation
    is the two-launch compact path; sampled windows against the oracle."""
    ev, offsets, gt = synth.make_stream(0, 24)
    with ctx_for(ebo, synth, 0, tv_weight=0.0, max_events=len(ev), max_windows=24) as c:
        c.set_windows(ev, offsets)
        flows = gt * 0.7
        r, J = c.eval(flows)
        for w in (0, 17):
            ro, Jo, _, _ = orc.window_eval(ev[offsets[w]:offsets[w + 1]], oparams(orc, c.params), flows[w])
            check_rj(r[w], J[w], ro, Jo)


@pytest.mark.parametrize("config,windows", [(0, 2), (3, 1)])
def test_edge_image_rows_stored_for_the_taps_only(ebo, orc, synth, config, windows):
    """Round 5: I is stored for the rows that hold taps plus one zero row on either side, every stencil read clamps its
    row into them.  Uniform flows that push every patch's events against each border of its 3W x 3H canvas (the stored
    rows then end at the canvas edge, where there is no zero row) and a diagonal one, value and Jacobian against the
    oracle; the value-only launch gives the same value bits."""
    ev, offsets, gt = synth.make_stream(config, windows)
    with ctx_for(ebo, synth, config, tv_weight=0.0, max_events=len(ev), max_windows=windows) as c:
        c.set_windows(ev, offsets)
        prm = oparams(orc, c.params)
        for m in ((0.9, 0.0), (-0.9, 0.0), (0.0, 0.9), (0.0, -0.9), (0.8, -0.8), (0.05, 0.02)):
            flows = np.tile(np.array(m), (windows, c.P, 1)) + 0.3 * gt
            r, J = c.eval(flows)
            v, _ = c.eval(flows, want_jac=False)
            assert np.array_equal(r, v)
            for w in range(windows):
                ro, Jo, _, _ = orc.window_eval(ev[offsets[w]:offsets[w + 1]], prm, flows[w])
                check_rj(r[w], J[w], ro, Jo)
