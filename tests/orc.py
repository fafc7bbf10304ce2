"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("EBO_ORACLE_LIB", os.path.join(ORACLE_DIR, "liboracle.so"))

# numpy mirror of orc_event / ebo_event / common::EventSample (24 bytes)
EVENT_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("sign", "<i4"), ("reserved", "<i4"), ("t_us", "<i8")]
)
assert EVENT_DTYPE.itemsize == 24


class FunctorConsts(C.Structure):
    _fields_ = [
        ("max_possible_residual", C.c_double),
        ("sigma_compensate", C.c_double),
        ("kernel_compensate", C.c_int32),
        ("kernel_st", C.c_int32),
        ("sigma_st", C.c_double),
        ("kernel_nms", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("image_w", C.c_int32),
        ("image_h", C.c_int32),
        ("patch_w", C.c_int32),
        ("patch_h", C.c_int32),
        ("tv_weight", C.c_double),
        ("tv_huber", C.c_double),
        ("scale", C.c_double),
        ("min_events", C.c_uint32),
        ("loss", C.c_int32),
        ("k", FunctorConsts),
    ]


class SolverOpts(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("use_nonmonotonic", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("max_consecutive_nonmonotonic", C.c_int32),
        ("max_consecutive_invalid", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("mode", C.c_int32),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("num_evals_cost", C.c_int32),
        ("num_evals_jac", C.c_int32),
        ("termination", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_mid_timestamp.restype = C.c_int64
        _lib.orc_mid_timestamp.argtypes = [C.c_int64, C.c_int64]
    return _lib


def _evp(ev):
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    return ev, ev.ctypes.data_as(C.c_void_p)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_events(x, y, t_us, sign=None):
    n = len(x)
    ev = np.zeros(n, dtype=EVENT_DTYPE)
    ev["x"] = x
    ev["y"] = y
    ev["t_us"] = t_us
    ev["sign"] = 1 if sign is None else sign
    return ev


def default_consts():
    k = FunctorConsts()
    lib().orc_default_consts(C.byref(k))
    return k


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for key, val in kw.items():
        setattr(p, key, val)
    return p


def default_solver(**kw):
    o = SolverOpts()
    lib().orc_default_solver(C.byref(o))
    for key, val in kw.items():
        setattr(o, key, val)
    return o


def contrast_eval(ev, rect, motion, loss, want_jac=True, scale=1e-3, consts=None):
    """contrastFunctor::operator() on one patch. Returns (r, J or None)."""
    ev, p = _evp(ev)
    k = consts or default_consts()
    m = np.asarray(motion, dtype=np.float64)
    r = C.c_double()
    J = np.zeros(2)
    rc = lib().orc_contrast_eval(
        p, C.c_size_t(len(ev)), *[int(v) for v in rect], C.c_double(scale), C.byref(k),
        int(loss), _dp(m), C.byref(r), _dp(J) if want_jac else None)
    assert rc == 0
    return r.value, (J if want_jac else None)


def contrast_image(ev, rect, motion, channels=3, scale=1e-3, consts=None):
    ev, p = _evp(ev)
    k = consts or default_consts()
    m = np.asarray(motion, dtype=np.float64)
    img = np.zeros((channels, 3 * rect[3], 3 * rect[2]))
    rc = lib().orc_contrast_image(
        p, C.c_size_t(len(ev)), *[int(v) for v in rect], C.c_double(scale), C.byref(k),
        _dp(m), int(channels), _dp(img))
    assert rc == 0
    return img


def tv_eval(weight, x, y):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    r = np.zeros(2)
    jx = np.zeros((2, 2))
    jy = np.zeros((2, 2))
    rc = lib().orc_tv_eval(C.c_double(weight), _dp(x), _dp(y), _dp(r), _dp(jx), _dp(jy))
    assert rc == 0
    return r, jx, jy


def grid(prm):
    npx, npy = C.c_int(), C.c_int()
    assert lib().orc_grid(C.byref(prm), C.byref(npx), C.byref(npy)) == 0
    return npx.value, npy.value


def patch_rect(prm, px, py):
    v = [C.c_int() for _ in range(4)]
    assert lib().orc_patch_rect(C.byref(prm), px, py, *[C.byref(a) for a in v]) == 0
    return tuple(a.value for a in v)


def window_eval(ev, prm, flows, want_jac=True):
    ev, p = _evp(ev)
    npx, npy = grid(prm)
    P = npx * npy
    flows = np.ascontiguousarray(flows, dtype=np.float64).reshape(P, 2)
    r = np.zeros(P)
    J = np.zeros((P, 2))
    active = np.zeros(P, dtype=np.int32)
    counts = np.zeros(P, dtype=np.int32)
    rc = lib().orc_window_eval(
        p, C.c_size_t(len(ev)), C.byref(prm), _dp(flows), _dp(r),
        _dp(J) if want_jac else None, active.ctypes.data_as(C.c_void_p),
        counts.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return r, (J if want_jac else None), active, counts


def window_eval_timed(ev, prm, flows, want_jac=True, reps=1):
    """(seconds, event_evaluations) of `reps` batched evaluations on one core."""
    ev, p = _evp(ev)
    flows = np.ascontiguousarray(flows, dtype=np.float64)
    sec = C.c_double()
    cnt = C.c_uint64()
    rc = lib().orc_window_eval_timed(
        p, C.c_size_t(len(ev)), C.byref(prm), _dp(flows), int(bool(want_jac)), int(reps),
        C.byref(sec), C.byref(cnt))
    assert rc == 0
    return sec.value, cnt.value


def compensate_events_contrast(ev, prm, opts=None, want_image=True):
    ev, p = _evp(ev)
    opts = opts or default_solver()
    npx, npy = grid(prm)
    flows = np.zeros((npx * npy, 2))
    img = np.zeros((prm.image_h, prm.image_w)) if want_image else None
    s = Summary()
    rc = lib().orc_compensate_events_contrast(
        p, C.c_size_t(len(ev)), C.byref(prm), C.byref(opts), _dp(flows),
        _dp(img) if want_image else None, C.byref(s))
    assert rc == 0
    return flows, img, s


def final_count_image(ev, prm, flows):
    ev, p = _evp(ev)
    flows = np.ascontiguousarray(flows, dtype=np.float64)
    img = np.zeros((prm.image_h, prm.image_w))
    rc = lib().orc_final_count_image(p, C.c_size_t(len(ev)), C.byref(prm), _dp(flows), _dp(img))
    assert rc == 0
    return img


def integrate_events(ev, w, h):
    ev, p = _evp(ev)
    img = np.zeros((h, w))
    assert lib().orc_integrate_events(p, C.c_size_t(len(ev)), int(w), int(h), _dp(img)) == 0
    return img


def compensate_events_field(ev, w, h, field, scale=1e-3):
    ev, p = _evp(ev)
    field = np.ascontiguousarray(field, dtype=np.float32).reshape(h, w, 2)
    img = np.zeros((h, w))
    rc = lib().orc_compensate_events_field(
        p, C.c_size_t(len(ev)), int(w), int(h), C.c_double(scale),
        field.ctypes.data_as(C.c_void_p), _dp(img))
    assert rc == 0
    return img


def patch_integrate(ev, rect):
    """Patch::integrateEvents; ev in deque order (front = newest)."""
    ev, p = _evp(ev)
    rx, ry, rw, rh = [float(v) for v in rect]
    nabla = np.zeros((int(rh), int(rw)))
    cur, last = C.c_int64(), C.c_int64()
    rc = lib().orc_patch_integrate(
        p, C.c_size_t(len(ev)), C.c_double(rx), C.c_double(ry), C.c_double(rw),
        C.c_double(rh), _dp(nabla), C.byref(cur), C.byref(last))
    assert rc == 0
    return nabla, cur.value, last.value


def route_events(ev, rects, start, max_take, cap):
    """FeatureDetector::updatePatches' routing loop -> (index arrays per patch, next per patch)."""
    ev, p = _evp(ev)
    rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
    n = len(rects)
    start = np.ascontiguousarray(start, dtype=np.uint32)
    max_take = np.ascontiguousarray(max_take, dtype=np.uint32)
    idx = np.zeros((n, max(int(cap), 1)), dtype=np.uint32)
    cnt = np.zeros(n, dtype=np.uint32)
    nxt = np.zeros(n, dtype=np.uint32)
    rc = lib().orc_route_events(p, C.c_size_t(len(ev)), n, _dp(rects), start.ctypes.data_as(C.c_void_p),
                                max_take.ctypes.data_as(C.c_void_p), C.c_uint32(int(cap)),
                                idx.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
                                nxt.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return [idx[i, :cnt[i]].copy() for i in range(n)], nxt


def patch_integrate_mc(ev, rect, prelast, last, mid_time):
    """Patch::integrateMotionCompensatedEvents; prelast/last = (x, y, t_us)."""
    ev, p = _evp(ev)
    rx, ry, rw, rh = [float(v) for v in rect]
    nabla = np.zeros((int(rh), int(rw)))
    a = np.asarray(prelast[:2], dtype=np.float64)
    b = np.asarray(last[:2], dtype=np.float64)
    upd = C.c_int32()
    rc = lib().orc_patch_integrate_mc(
        p, C.c_size_t(len(ev)), C.c_double(rx), C.c_double(ry), C.c_double(rw),
        C.c_double(rh), _dp(a), C.c_int64(int(prelast[2])), _dp(b), C.c_int64(int(last[2])),
        C.c_int64(int(mid_time)), _dp(nabla), C.byref(upd))
    assert rc == 0
    return nabla, bool(upd.value)


def parse_events_txt(path, cap=1 << 20):
    out = np.zeros(cap, dtype=EVENT_DTYPE)
    n = C.c_size_t()
    rc = lib().orc_parse_events_txt(
        path.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n))
    return rc, out[: n.value].copy()


def mid_timestamp(front, back):
    return lib().orc_mid_timestamp(int(front), int(back))


def init_motion_field(w, h, timestamp, trajectories, use_average=True, scale=1e-3):
    """FeatureDetector::initMotionField. trajectories: list of [(x, y, t_us), ...] per patch."""
    offs = np.zeros(len(trajectories) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(t) for t in trajectories])
    flat = [s for t in trajectories for s in t]
    xy = np.ascontiguousarray([[s[0], s[1]] for s in flat], dtype=np.float64).reshape(-1, 2)
    tt = np.ascontiguousarray([int(s[2]) for s in flat], dtype=np.int64)
    field = np.zeros((h, w, 2), dtype=np.float32)
    nfix = C.c_int32()
    fixed = np.zeros((max(len(trajectories), 1), 2), dtype=np.int32)
    rc = lib().orc_init_motion_field(
        int(w), int(h), C.c_double(scale), int(bool(use_average)), len(trajectories),
        offs.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p),
        C.c_int64(int(timestamp)), field.ctypes.data_as(C.c_void_p), C.byref(nfix),
        fixed.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return field, fixed[: nfix.value].copy()


def interpolate_motion_field(field, fixed, use_l1=False, opts=None):
    """FeatureDetector::interpolateMotionField after its initMotionField call.
    field float32 [h][w][2] (copied), fixed int32 [n][2] (x, y).  Returns (field, summary, rc)."""
    out = np.ascontiguousarray(field, dtype=np.float32).copy()
    h, w = out.shape[:2]
    fx = np.ascontiguousarray(fixed, dtype=np.int32).reshape(-1, 2)
    s = Summary()
    rc = lib().orc_interpolate_motion_field(
        int(w), int(h), int(bool(use_l1)), out.ctypes.data_as(C.c_void_p), len(fx),
        fx.ctypes.data_as(C.c_void_p), C.byref(opts) if opts is not None else None, C.byref(s))
    return out, s, rc


# ---- per-feature tracker objective (optimizer_oracle.cpp) -------------------------------
def optimizer_cost(grad, rect, nabla, pose, flow_dir, want_jac=True):
    """OptimizerCostFunctor::operator().  grad [H][W][2], rect (x, y, w, h) doubles,
    nabla [h][w].  Returns (residuals [n], jac_pose [n][4] | None, jac_flow [n] | None)."""
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    nabla = np.ascontiguousarray(nabla, dtype=np.float64)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    h, w = grad.shape[:2]
    n = int(rect[2]) * int(rect[3])
    res = np.zeros(n)
    jp = np.zeros((n, 4)) if want_jac else None
    jf = np.zeros(n) if want_jac else None
    rc = lib().orc_optimizer_cost(
        _dp(grad), w, h, C.c_double(rect[0]), C.c_double(rect[1]), C.c_double(rect[2]), C.c_double(rect[3]),
        _dp(nabla), _dp(pose), C.c_double(flow_dir), _dp(res), _dp(jp) if want_jac else None,
        _dp(jf) if want_jac else None)
    assert rc == 0
    return res, jp, jf


def optimizer_cost_map(grad, rect, nabla, pose, flow_dir, map_w=11, map_h=11):
    """Optimizer::drawCostMap (optimizer.cpp:33-60) -> [map_h][map_w]."""
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    nabla = np.ascontiguousarray(nabla, dtype=np.float64)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    h, w = grad.shape[:2]
    out = np.zeros((map_h, map_w))
    rc = lib().orc_optimizer_cost_map(
        _dp(grad), w, h, C.c_double(rect[0]), C.c_double(rect[1]), C.c_double(rect[2]), C.c_double(rect[3]),
        _dp(nabla), _dp(pose), C.c_double(flow_dir), int(map_w), int(map_h), _dp(out))
    assert rc == 0
    return out


def optimizer_default_solver(**kw):
    o = SolverOpts()
    lib().orc_optimizer_default_solver(C.byref(o))
    for key, val in kw.items():
        setattr(o, key, val)
    return o


def optimizer_solve(grad, rect, nabla, pose, flow_dir, huber=0.3, opts=None):
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    nabla = np.ascontiguousarray(nabla, dtype=np.float64)
    p = np.ascontiguousarray(pose, dtype=np.float64).copy()
    fd = C.c_double(flow_dir)
    h, w = grad.shape[:2]
    s = Summary()
    rc = lib().orc_optimizer_solve(
        _dp(grad), w, h, C.c_double(rect[0]), C.c_double(rect[1]), C.c_double(rect[2]), C.c_double(rect[3]),
        _dp(nabla), C.c_double(huber), C.byref(opts) if opts is not None else None, _dp(p), C.byref(fd),
        C.byref(s))
    assert rc == 0
    return p, fd.value, s


def patch_warp_image(grad, rect, pose, flow_dir):
    """Patch::warpImage (patch.cpp:132-154) -> (predictedNabla [h][w] or None on the border return)"""
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    h, w = grad.shape[:2]
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    ph, pw = int(np.rint(rect[3])), int(np.rint(rect[2]))
    out = np.zeros((ph, pw))
    upd = C.c_int()
    rc = lib().orc_patch_warp_image(_dp(grad), w, h, C.c_double(rect[0]), C.c_double(rect[1]), C.c_double(rect[2]),
                                    C.c_double(rect[3]), _dp(pose), C.c_double(flow_dir), _dp(out), C.byref(upd))
    assert rc == 0
    return out if upd.value else None


def estimate_num_events(grad, rect, pose, flow_dir):
    """The event-count estimate of FeatureDetector::updateNumOfEvents (feature_detector.cpp:689-707)."""
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    h, w = grad.shape[:2]
    out = C.c_uint64()
    rc = lib().orc_estimate_num_events(_dp(grad), w, h, C.c_double(rect[0]), C.c_double(rect[1]), C.c_double(rect[2]),
                                       C.c_double(rect[3]), _dp(pose), C.c_double(flow_dir), C.byref(out))
    assert rc == 0
    return out.value


def se2_plus(pose, delta3):
    out = np.zeros(4)
    assert lib().orc_se2_plus(_dp(np.ascontiguousarray(pose, dtype=np.float64)),
                              _dp(np.ascontiguousarray(delta3, dtype=np.float64)), _dp(out)) == 0
    return out


def patch_update_rect(warp, init_xy, rw, rh):
    out = np.zeros(4)
    assert lib().orc_patch_update_rect(_dp(np.ascontiguousarray(warp, dtype=np.float64)),
                                       C.c_double(init_xy[0]), C.c_double(init_xy[1]),
                                       C.c_double(rw), C.c_double(rh), _dp(out)) == 0
    return out


def normalize_nabla(nabla):
    a = np.ascontiguousarray(nabla, dtype=np.float64)
    out = np.zeros_like(a)
    assert lib().orc_normalize_nabla(_dp(a), a.size, _dp(out)) == 0
    return out


def lm_powell(x0=(3.0, -1.0, 0.0, 1.0), opts=None, cap=200):
    """oracle.cpp::minimize on Powell's function.  Returns (x, trace [n][6], summary)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    trace = np.zeros((cap, 6))
    n = C.c_int32()
    s = Summary()
    rc = lib().orc_lm_powell(C.byref(opts) if opts is not None else None, _dp(x), _dp(trace), cap,
                             C.byref(n), C.byref(s))
    assert rc == 0
    return x, trace[: n.value].copy(), s
