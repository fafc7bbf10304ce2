"""FeatureDetector::interpolateMotionField (feature_detector.cpp:144-241): the per-pixel TV
smoothing of the motion field.

CPU: the oracle against an independent replay of the same Levenberg-Marquardt iteration with
scipy's sparse direct solver (another elimination order, as Ceres' own is), and the semantics
the reference's code implies (unused last pixel, leaf row/column, cv::norm quirk, constants).
GPU: the device solve (conjugate gradients inside the LM loop) against the oracle.

Tolerances: the field is stored as float32 (:230-239).  Two correct double-precision solves of
the same LM iteration differ by ~1e-12, which rounds to the same float32 except at a rounding
boundary; the test allows 1 float32 ulp at <= 0.1 % of the pixels and 2e-7 absolute elsewhere.
"""
import numpy as np
import pytest


def make_case(orc, w, h, n_patches, seed, use_average=True):
    rng = np.random.default_rng(seed)
    traj = []
    for _ in range(n_patches):
        v = rng.uniform(-1, 1, 2) * 1e-3  # px per us
        # lower_bound(25000) finds the sample at t = 31000: keep that point inside the image
        x0, y0 = rng.uniform(2, w - 3) - v[0] * 30000, rng.uniform(2, h - 3) - v[1] * 30000
        traj.append([(x0 + v[0] * t, y0 + v[1] * t, 1000 + t) for t in range(0, 60000, 10000)])
    field, fixed = orc.init_motion_field(w, h, 25000, traj, use_average=use_average)
    return traj, field, fixed


def lm_replay_scipy(field, fixed, iters_cap=50, ftol=1e-6, gtol=1e-10, ptol=1e-8):
    """Ceres' trust-region LM on the quadratic (no loss) TV problem, normal equations solved by
    scipy.sparse.linalg.spsolve.  Independent of the oracle's assembly and banded Cholesky."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    h, w = field.shape[:2]
    n = w * h
    fx = np.zeros(n, bool)
    fx[fixed[:, 1] * w + fixed[:, 0]] = True
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    own = ((xs <= w - 2) & (ys <= h - 2)).ravel()
    p = np.arange(n)[own]
    pe = np.concatenate([p, p])
    qe = np.concatenate([p + 1, p + w])
    keep = ~(fx[pe] & fx[qe])
    pe, qe = pe[keep], qe[keep]
    ne = len(pe)
    B = sp.coo_matrix((np.r_[np.ones(ne), -np.ones(ne)], (np.r_[np.arange(ne), np.arange(ne)], np.r_[pe, qe])),
                      shape=(ne, n)).tocsr()
    L = (B.T @ B).tocsr()
    deg = L.diagonal()
    free = (deg > 0) & ~fx
    fi = np.where(free)[0]
    Lff = L[fi][:, fi].tocsc()
    x = field.reshape(n, 2).astype(np.float64)
    s2 = 1.0 / (1.0 + np.sqrt(deg[fi])) ** 2

    def cost(xx):
        d = B @ xx
        return 0.5 * float((d * d).sum())

    c = cost(x)
    radius, it = 1e4, 0
    last_ok = True
    while True:
        g = (L @ x)[fi]
        if it >= iters_cap or (last_ok and np.abs(g).max() <= gtol):
            break
        it += 1
        last_ok = False
        dmp = np.clip(s2 * deg[fi], 1e-6, 1e32) / radius / s2
        y = spl.spsolve(Lff + sp.diags(dmp).tocsc(), -g)
        model = -float((y * g).sum()) - 0.5 * float((y * (Lff @ y)).sum())
        assert model > 0
        xc = x.copy()
        xc[fi] += y
        cc = cost(xc)
        if np.sqrt((y * y).sum()) <= ptol * (np.sqrt((x[fi] ** 2).sum()) + ptol):
            break
        if abs(c - cc) <= ftol * c:
            break
        q = (c - cc) / model
        assert q > 1e-3  # a quadratic: every step is accepted
        x, c, last_ok = xc, cc, True
        radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * q - 1.0) ** 3))
    return x.reshape(h, w, 2).astype(np.float32), it, c


def ulp_report(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
    ulp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
    return diff, float((diff > 0).mean()), bool(np.all(diff <= np.maximum(1.01 * ulp, 2e-7)))


@pytest.mark.parametrize("w,h,npatch,seed", [(40, 30, 6, 1), (64, 48, 10, 2), (33, 57, 4, 3)])
def test_oracle_field_tv_matches_independent_lm_replay(orc, w, h, npatch, seed):
    _, field, fixed = make_case(orc, w, h, npatch, seed)
    assert len(fixed) >= 2
    out, s, rc = orc.interpolate_motion_field(field, fixed)
    assert rc == 0 and s.termination == 0
    ref, it, c = lm_replay_scipy(field, fixed)
    assert s.iterations == it
    assert s.final_cost == pytest.approx(c, rel=1e-10)
    diff, frac, ok = ulp_report(out, ref)
    assert ok and frac <= 1e-3, (diff.max(), frac)


def test_oracle_field_tv_semantics(orc):
    w, h = 24, 16
    _, field, fixed = make_case(orc, w, h, 5, 7)
    out, s, rc = orc.interpolate_motion_field(field, fixed)
    assert rc == 0 and s.iterations >= 1 and s.final_cost < s.initial_cost
    # constants stay (:206-214); pixel (w-1, h-1) is in no residual block (:170-204)
    for x, y in fixed:
        assert np.array_equal(out[y, x], field[y, x])
    assert np.array_equal(out[h - 1, w - 1], field[h - 1, w - 1])
    # maximum principle of the smoothing: values stay inside the range of the initial field
    for c in range(2):
        assert out[..., c].min() >= field[..., c].min() - 1e-6
        assert out[..., c].max() <= field[..., c].max() + 1e-6
    # the last column / row hang on one edge only: they follow their single neighbour
    assert np.abs(out[: h - 1, w - 1] - out[: h - 1, w - 2]).max() < 1e-2
    assert np.abs(out[h - 1, : w - 1] - out[h - 2, : w - 1]).max() < 1e-2
    # a fixed point at the unused pixel is Ceres' abort in the reference (:208)
    bad = np.vstack([fixed, [[w - 1, h - 1]]]).astype(np.int32)
    assert orc.interpolate_motion_field(field, bad)[2] == -2
    # cv::norm(motionField_) > 0 (:152) reads float pairs as doubles: with every second
    # component zero the doubles are subnormal, their squares vanish, nothing is smoothed
    f2 = field.copy()
    f2[..., 1] = 0
    out2, s2, _ = orc.interpolate_motion_field(f2, fixed)
    assert s2.iterations == 0 and np.array_equal(out2, f2)
    zero = np.zeros_like(field)
    out3, s3, _ = orc.interpolate_motion_field(zero, fixed)
    assert s3.iterations == 0 and not out3.any()


def test_oracle_field_tv_l1_runs_and_reduces_huber_cost(orc):
    _, field, fixed = make_case(orc, 32, 24, 6, 11)
    out, s, rc = orc.interpolate_motion_field(field, fixed, use_l1=True)
    assert rc == 0 and s.termination in (0, 1)
    assert s.final_cost < s.initial_cost
    for x, y in fixed:
        assert np.array_equal(out[y, x], field[y, x])


# ------------------------------------------------------------------ GPU
def run_device(ebo, orc, w, h, traj, use_average, use_l1, opts=None):
    p = ebo.default_params()
    p.image_w, p.image_h = w, h
    p.patch_w, p.patch_h = min(20, w), min(20, h)
    c = ebo.Context(p)
    try:
        field, fixed = c.init_motion_field(25000, traj, use_average=use_average)
        out, s, cg = c.interpolate_motion_field(use_l1=use_l1, opts=opts)
    finally:
        c.close()
    return field, fixed, out, s, cg


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,npatch,seed,avg", [(40, 30, 6, 1, True), (64, 48, 10, 2, True),
                                                 (33, 57, 4, 3, False), (96, 72, 12, 4, True)])
def test_device_field_tv_matches_oracle(ebo, orc, w, h, npatch, seed, avg):
    traj, field_o, fixed_o = make_case(orc, w, h, npatch, seed, use_average=avg)
    field, fixed, out, s, cg = run_device(ebo, orc, w, h, traj, avg, False)
    assert np.array_equal(field, field_o) and np.array_equal(fixed, fixed_o)
    ref, so, rc = orc.interpolate_motion_field(field_o, fixed_o)
    assert rc == 0
    assert (s.iterations, s.termination) == (so.iterations, so.termination)
    assert s.initial_cost == pytest.approx(so.initial_cost, rel=1e-12)
    assert s.final_cost == pytest.approx(so.final_cost, rel=1e-10)
    diff, frac, ok = ulp_report(out, ref)
    assert ok and frac <= 1e-3, (diff.max(), frac)
    assert cg > 0


@pytest.mark.gpu
def test_device_field_tv_l1_matches_oracle(ebo, orc):
    w, h = 48, 36
    traj, field_o, fixed_o = make_case(orc, w, h, 8, 5)
    opts = ebo.default_solver()
    opts.use_nonmonotonic = 0
    opts.function_tolerance, opts.gradient_tolerance, opts.parameter_tolerance = 1e-6, 1e-10, 1e-8
    opts.max_num_iterations = 6  # IRLS on |.|: compare a fixed number of iterations
    oo = orc.default_solver(use_nonmonotonic=0, function_tolerance=1e-6, gradient_tolerance=1e-10,
                            parameter_tolerance=1e-8, max_num_iterations=6)
    _, _, out, s, _ = run_device(ebo, orc, w, h, traj, True, True, opts)
    ref, so, rc = orc.interpolate_motion_field(field_o, fixed_o, use_l1=True, opts=oo)
    assert rc == 0 and s.iterations == so.iterations
    assert s.final_cost == pytest.approx(so.final_cost, rel=1e-8)
    assert np.abs(out.astype(np.float64) - ref).max() < 1e-5


@pytest.mark.gpu
def test_device_field_tv_edge_cases(ebo, orc):
    w, h = 24, 16
    p = ebo.default_params()
    p.image_w, p.image_h, p.patch_w, p.patch_h = w, h, 12, 8
    c = ebo.Context(p)
    try:
        with pytest.raises(ebo.EboError):
            c.interpolate_motion_field()  # no field yet
        # nothing fixed -> zero field -> norm 0 -> untouched
        f, fx = c.init_motion_field(0, [[(3.0, 3.0, 100), (4.0, 3.0, 1100)]])
        out, s, cg = c.interpolate_motion_field()
        assert not out.any() and s.iterations == 0 and cg == 0
        # second component zero everywhere: the reinterpretation quirk leaves the field alone
        f, fx = c.init_motion_field(50, [[(3.0, 3.0, 100), (4.0, 3.0, 1100)], [(15.0, 9.0, 100), (17.0, 9.0, 1100)]])
        assert len(fx) == 2 and not f[..., 1].any()
        out, s, cg = c.interpolate_motion_field()
        assert np.array_equal(out, f) and s.iterations == 0
        # fixed point at the last pixel: the reference aborts inside Ceres
        f, fx = c.init_motion_field(50, [[(w - 1.0, h - 1.0, 100), (w - 2.0, h - 2.0, 1100)]])
        assert fx.tolist() == [[w - 1, h - 1]]
        with pytest.raises(ebo.EboError):
            c.interpolate_motion_field()
    finally:
        c.close()


@pytest.mark.gpu
def test_device_compensate_events_with_smoothed_field(ebo, orc, synth):
    """compensateEvents with optimizeFlowTV (feature_detector.cpp:243-296): the count image
    through the smoothed field equals the oracle's through its own smoothed field."""
    w, h = 240, 180
    traj, field_o, fixed_o = make_case(orc, w, h, 40, 9)
    ev, _ = synth.make_window(2, n_events=20000, event_dtype=ebo.EVENT_DTYPE)
    p = ebo.default_params()
    p.image_w, p.image_h = w, h
    c = ebo.Context(p)
    try:
        c.init_motion_field(25000, traj)
        out, s, cg = c.interpolate_motion_field()
        c.set_window(ev)
        img = c.count_image(ebo.COUNT_FIELD, out)[0]
    finally:
        c.close()
    ref, so, _ = orc.interpolate_motion_field(field_o, fixed_o)
    assert s.iterations == so.iterations
    diff, frac, ok = ulp_report(out, ref)
    assert ok and frac <= 1e-3, (diff.max(), frac)
    img_same_field = orc.compensate_events_field(ev, w, h, out)
    assert np.array_equal(img, img_same_field)  # bit-exact given the same field
    img_o = orc.compensate_events_field(ev, w, h, ref)
    assert np.abs(img - img_o).sum() <= 4  # an ulp of the field can move an event on a rounding boundary


@pytest.mark.gpu
def test_device_field_tv_diagonal_preconditioner_agrees(ebo_ab, orc, monkeypatch):
    ebo = ebo_ab  # libebo_hip_ab.so: the build that reads the EBO_* switches (csrc/ab_env.h)
    """EBO_TVF_PRECOND=jacobi (the fallback) and the multigrid preconditioner solve the same
    systems: same LM trajectory, same field, ~20x apart in CG iterations."""
    w, h = 96, 72
    traj, field_o, fixed_o = make_case(orc, w, h, 12, 4)
    _, _, out_mg, s_mg, cg_mg = run_device(ebo, orc, w, h, traj, True, False)
    monkeypatch.setenv("EBO_TVF_PRECOND", "jacobi")
    _, _, out_j, s_j, cg_j = run_device(ebo, orc, w, h, traj, True, False)
    assert (s_mg.iterations, s_mg.termination) == (s_j.iterations, s_j.termination)
    assert s_mg.final_cost == pytest.approx(s_j.final_cost, rel=1e-10)
    diff, frac, ok = ulp_report(out_mg, out_j)
    assert ok and frac <= 1e-3
    assert cg_j > 5 * cg_mg


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_device_field_tv_random_sweep(ebo, orc, seed):
    """Random image sizes (odd, non-multiples of the aggregate size), 2..20 tracked points, both
    fill modes, with and without the Huber loss."""
    rng = np.random.default_rng(100 + seed)
    w, h = int(rng.integers(17, 110)), int(rng.integers(13, 90))
    npatch = int(rng.integers(2, 21))
    avg = bool(seed % 2)
    use_l1 = seed % 4 == 3
    traj, field_o, fixed_o = make_case(orc, w, h, npatch, 300 + seed, use_average=avg)
    if len(fixed_o) < 2:
        pytest.skip("fewer than two fixed points")
    opts = oo = None
    if use_l1:
        opts = ebo.default_solver()
        opts.use_nonmonotonic = 0
        opts.function_tolerance, opts.gradient_tolerance, opts.parameter_tolerance = 1e-6, 1e-10, 1e-8
        opts.max_num_iterations = 5
        oo = orc.default_solver(use_nonmonotonic=0, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                parameter_tolerance=1e-8, max_num_iterations=5)
    field, fixed, out, s, cg = run_device(ebo, orc, w, h, traj, avg, use_l1, opts)
    assert np.array_equal(field, field_o) and np.array_equal(fixed, fixed_o)
    ref, so, rc = orc.interpolate_motion_field(field_o, fixed_o, use_l1=use_l1, opts=oo)
    assert rc == 0 and s.iterations == so.iterations
    assert s.final_cost == pytest.approx(so.final_cost, rel=1e-8)
    if use_l1:
        assert np.abs(out.astype(np.float64) - ref).max() < 1e-5
    else:
        diff, frac, ok = ulp_report(out, ref)
        assert ok and frac <= 2e-3, (diff.max(), frac)
