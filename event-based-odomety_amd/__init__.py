"""event-based-odomety_amd — MI355X (gfx950) implementation of the motion-compensated
event-warping path of nurlanov-zh/event-based-odomety.

The product is the C-ABI library ``libebo_hip.so`` (include/ebo.h) plus the C++
façade in ``include/feature_tracker``.  This module is only the ctypes plumbing the
Python tests and bench.py use to call that ABI; it contains no compute and no CPU
fallback.  Importing it never needs a GPU; creating a Context does.

The directory name has a hyphen (the upstream repository name), so import it with
``importlib.import_module("event-based-odomety_amd")``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# EBO_LIB_PATH: another build of the same library (tools/ A/B runs of two builds on one box only)
LIB_PATH = os.environ.get("EBO_LIB_PATH") or os.path.join(HERE, "libebo_hip.so")
# The A/B build (make ab: -DEBO_AB, the environment switches of csrc/ab_env.h and the superseded kernels).  A module
# instance imported under the name `..._ab` (tests' `ebo_ab` fixture, tools/ab/*) binds it; the product never does.
AB_LIB_PATH = os.path.join(HERE, "libebo_hip_ab.so")
if __name__.endswith("_ab"):
    LIB_PATH = AB_LIB_PATH
HEADER_PATH = os.path.join(ROOT, "include", "ebo.h")

OK = 0
ERR_ARG, ERR_HIP, ERR_RANGE, ERR_STATE, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_SOLVER, ERR_COMM = (
    -1, -2, -3, -4, -5, -6, -7, -8)
LOSS_EDGE, LOSS_VARIANCE = 0, 1
GRAD_JET, GRAD_CENTRAL = 0, 1
SOLVE_GLOBAL, SOLVE_INDEPENDENT = 0, 1
COUNT_INTEGRATED, COUNT_WARPED, COUNT_FIELD = 0, 1, 2

# numpy mirror of ebo_event (== common::EventSample, 24 bytes)
EVENT_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("sign", "<i4"), ("reserved", "<i4"), ("t_us", "<i8")]
)

# numpy mirror of ebo_event8 (compact raw event: x:15 | polarity:1 | y:15 | 0:1, int32 us relative to a base time)
EVENT8_DTYPE = np.dtype([("xy", "<u4"), ("t_rel_us", "<i4")])

# numpy mirror of ebo_track_point (one "feature_id timestamp x y" record, 32 bytes)
TRACK_DTYPE = np.dtype([("id", "<i8"), ("t_us", "<i8"), ("x", "<f8"), ("y", "<f8")])


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
        ("device", C.c_int32),
        ("image_w", C.c_int32),
        ("image_h", C.c_int32),
        ("patch_w", C.c_int32),
        ("patch_h", C.c_int32),
        ("tv_weight", C.c_double),
        ("tv_huber", C.c_double),
        ("scale", C.c_double),
        ("min_events", C.c_uint32),
        ("loss", C.c_int32),
        ("grad", C.c_int32),
        ("reserved", C.c_int32),
        ("fd_step", C.c_double),
        ("k", FunctorConsts),
        ("max_events", C.c_uint64),
        ("max_windows", C.c_int32),
        ("reserved2", C.c_int32),
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


class EboError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ebo error %d: %s" % (code, msg))
        self.code = code


_lib = None


def build(verbose=False):
    """Compile libebo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def lib():
    """The loaded library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: run __graft_entry__.build() (hipcc). "
                "This package has no CPU path." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.ebo_version.restype = C.c_char_p
        _lib.ebo_last_error.restype = C.c_char_p
        _lib.ebo_last_error.argtypes = [C.c_void_p]
        _lib.ebo_destroy.restype = None
        _lib.ebo_destroy.argtypes = [C.c_void_p]
        _lib.ebo_graph_destroy.restype = None
        _lib.ebo_graph_destroy.argtypes = [C.c_void_p]
        _lib.ebo_graph_begin.argtypes = [C.c_void_p]
        _lib.ebo_graph_end.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ebo_graph_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.ebo_lm_destroy.restype = None
        _lib.ebo_lm_destroy.argtypes = [C.c_void_p]
        _lib.ebo_lm_request.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ebo_lm_supply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.ebo_lm_result.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def version():
    return lib().ebo_version().decode()


def device_count():
    n = C.c_int()
    lib().ebo_device_count(C.byref(n))
    return n.value


def default_params(**kw):
    p = Params()
    lib().ebo_default_params(C.byref(p))
    for key, val in kw.items():
        if not hasattr(p, key):
            raise AttributeError(key)
        setattr(p, key, val)
    return p


def default_solver(**kw):
    o = SolverOpts()
    lib().ebo_default_solver(C.byref(o))
    for key, val in kw.items():
        if not hasattr(o, key):
            raise AttributeError(key)
        setattr(o, key, val)
    return o


def optimizer_default_solver(**kw):
    """ceres::Solver::Options as Optimizer::optimize sets them (optimizer.cpp:103-112)."""
    o = SolverOpts()
    lib().ebo_optimizer_default_solver(C.byref(o))
    for key, val in kw.items():
        if not hasattr(o, key):
            raise AttributeError(key)
        setattr(o, key, val)
    return o


class HostSolver:
    """ebo_lm_*: the host LM of EBO_SOLVE_GLOBAL as a resumable state machine (request -> evaluate -> supply)."""

    DONE, NEED_JACOBIAN, NEED_VALUE = 0, 1, 2

    def __init__(self, npx, npy, active, tv_weight=1e3, tv_huber=10.0, opts=None):
        self.P = int(npx) * int(npy)
        active = np.ascontiguousarray(active, dtype=np.uint8).reshape(self.P)
        self._opts = opts if opts is not None else default_solver()
        h = C.c_void_p()
        rc = lib().ebo_lm_create(int(npx), int(npy), _vp(active), C.c_double(tv_weight), C.c_double(tv_huber),
                                 C.byref(self._opts), C.byref(h))
        if rc:
            raise EboError(rc, "ebo_lm_create")
        self._h = h

    def request(self):
        flows = np.zeros((self.P, 2))
        rc = lib().ebo_lm_request(self._h, _dp(flows))
        if rc < 0:
            raise EboError(rc, "ebo_lm_request")
        return rc, flows

    def supply(self, r, jac=None):
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(self.P)
        j = None if jac is None else np.ascontiguousarray(jac, dtype=np.float64).reshape(self.P, 2)
        rc = lib().ebo_lm_supply(self._h, _dp(r), _dp(j) if j is not None else None)
        if rc:
            raise EboError(rc, "ebo_lm_supply")

    def result(self):
        flows = np.zeros((self.P, 2))
        s = Summary()
        rc = lib().ebo_lm_result(self._h, _dp(flows), C.byref(s))
        if rc:
            raise EboError(rc, "ebo_lm_result")
        return flows, s

    def close(self):
        if self._h:
            lib().ebo_lm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def window_ref_time(t_first_us, t_last_us):
    """The reference time of a window from its first / last event time (feature_detector.cpp:305-306)."""
    out = C.c_int64()
    rc = lib().ebo_window_ref_time(C.c_int64(int(t_first_us)), C.c_int64(int(t_last_us)), C.byref(out))
    if rc:
        raise EboError(rc, "window mid-time outside int32 microseconds")
    return out.value


def shard_range(n_units, rank, world):
    b, e = C.c_int(), C.c_int()
    rc = lib().ebo_shard_range(int(n_units), int(rank), int(world), C.byref(b), C.byref(e))
    if rc:
        raise EboError(rc, "bad shard arguments")
    return b.value, e.value


class Band(C.Structure):
    """ebo_band: one rank's share of the final image of row-sharded windows."""
    _fields_ = [("band_row0", C.c_int), ("own_row0", C.c_int), ("own_row1", C.c_int), ("band_row1", C.c_int),
                ("recv_above", C.c_int), ("recv_below", C.c_int)]

    top_rows = property(lambda s: s.own_row0 - s.band_row0)
    own_rows = property(lambda s: s.own_row1 - s.own_row0)
    bottom_rows = property(lambda s: s.band_row1 - s.own_row1)


def band_plan(image_h, row_bounds, rank, halo):
    rb = np.ascontiguousarray(row_bounds, dtype=np.int32)
    out = Band()
    rc = lib().ebo_band_plan(int(image_h), rb.ctypes.data_as(C.c_void_p), len(rb) - 1, int(rank), int(halo), C.byref(out))
    if rc:
        raise EboError(rc, "no band plan: bad row bounds, or a halo that does not fit inside a neighbour's rows")
    return out


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the other ranks receive its bytes)."""
    buf = (C.c_char * 128)()
    rc = lib().ebo_comm_unique_id(C.byref(buf))
    if rc:
        raise EboError(rc, lib().ebo_last_error(None).decode())
    return bytes(buf)


def read_events_txt(path, cap=1 << 22):
    """DAVIS240C events.txt -> structured event array (host-side parser of the library)."""
    out = np.zeros(cap, dtype=EVENT_DTYPE)
    n = C.c_size_t()
    rc = lib().ebo_read_events_txt(str(path).encode(), _vp(out), C.c_size_t(cap), C.byref(n))
    if rc:
        raise EboError(rc, "cannot parse %s (parsed %d events before the error)" % (path, n.value))
    return out[: n.value].copy()


def read_events_txt_threads(path, cap, threads=0, offset=None, out=None):
    """ebo_read_events_txt_threads: at most cap events with `threads` host threads (0: EBO_HOST_THREADS or the
    machine's) -> (events, next offset or None, threads that parsed).  `out`: a caller-owned array to fill (timing)."""
    out = np.zeros(cap, dtype=EVENT_DTYPE) if out is None else out
    n, used = C.c_size_t(), C.c_int()
    off = C.c_uint64(int(offset)) if offset is not None else None
    f = lib().ebo_read_events_txt_threads
    f.restype = C.c_int
    rc = f(str(path).encode(), C.byref(off) if off is not None else None, _vp(out), C.c_size_t(cap), C.byref(n),
           C.c_int(int(threads)), C.byref(used))
    if rc:
        raise EboError(rc, "cannot parse %s (parsed %d events before the error)" % (path, n.value))
    return out[: n.value], (int(off.value) if off is not None else None), int(used.value)


def read_events_txt8(path, cap, threads=0, offset=None):
    """ebo_read_events_txt8: at most cap events as compact 8-byte records -> (records, base time in us, next offset or
    None); every record's t_rel_us is relative to the base = the first event's time stamp of this call."""
    out = np.zeros(cap, dtype=EVENT8_DTYPE)
    n, base = C.c_size_t(), C.c_int64()
    off = C.c_uint64(int(offset)) if offset is not None else None
    f = lib().ebo_read_events_txt8
    f.restype = C.c_int
    rc = f(str(path).encode(), C.byref(off) if off is not None else None, _vp(out), C.c_size_t(cap), C.byref(n), C.byref(base),
           C.c_int(int(threads)))
    if rc:
        raise EboError(rc, "cannot parse %s into compact records (%d events before the error)" % (path, n.value))
    return out[: n.value], int(base.value), (int(off.value) if off is not None else None)


def read_events_txt_at(path, offset, cap=1_000_000):
    """At most cap events of an events.txt from byte `offset` on -> (events, next offset): the pieces
    Davis240cReader::getEvents reads a recording in (EVENT_LENGTH lines per call)."""
    out = np.zeros(cap, dtype=EVENT_DTYPE)
    n = C.c_size_t()
    off = C.c_uint64(int(offset))
    rc = lib().ebo_read_events_txt_at(str(path).encode(), C.byref(off), _vp(out), C.c_size_t(cap), C.byref(n))
    if rc:
        raise EboError(rc, "cannot parse %s (parsed %d events before the error)" % (path, n.value))
    return out[: n.value].copy(), int(off.value)


def write_events_bin(path, ev):
    """Packed binary sidecar (32-byte header + 16 B per event) of an event array."""
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    rc = lib().ebo_write_events_bin(str(path).encode(), _vp(ev), C.c_size_t(len(ev)))
    if rc:
        raise EboError(rc, "cannot write %s" % path)


def read_events_bin(path, cap=1 << 24):
    out = np.zeros(cap, dtype=EVENT_DTYPE)
    n = C.c_size_t()
    rc = lib().ebo_read_events_bin(str(path).encode(), _vp(out), C.c_size_t(cap), C.byref(n))
    if rc:
        raise EboError(rc, "cannot read %s (read %d events before the error)" % (path, n.value))
    return out[: n.value].copy()


def write_tracks_txt(path, pts):
    """trajectory.txt as tools::Evaluator::saveFeaturesTrajectory writes it (evaluator.cpp:125-150)."""
    pts = np.ascontiguousarray(pts, dtype=TRACK_DTYPE)
    rc = lib().ebo_write_tracks_txt(str(path).encode(), _vp(pts), C.c_size_t(len(pts)))
    if rc:
        raise EboError(rc, "cannot write %s" % path)


def read_tracks_txt(path, cap=1 << 16):
    """trajectory.txt -> track records; the buffer grows to what the file holds (never a truncated list)."""
    while True:
        out = np.zeros(cap, dtype=TRACK_DTYPE)
        n = C.c_size_t()
        rc = lib().ebo_read_tracks_txt(str(path).encode(), _vp(out), C.c_size_t(cap), C.byref(n))
        if rc == ERR_ARG and n.value > cap:
            cap = n.value
            continue
        if rc:
            raise EboError(rc, "cannot parse %s (parsed %d records before the error)" % (path, n.value))
        return out[: n.value].copy()


def pack_events8(ev, t_base, out=None):
    """EventSamples of one window -> compact 8-byte records relative to t_base."""
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    if out is None:
        out = np.zeros(len(ev), dtype=EVENT8_DTYPE)
    rc = lib().ebo_pack_events8(_vp(ev), C.c_size_t(len(ev)), C.c_int64(int(t_base)), _vp(out))
    if rc:
        raise EboError(rc, "event does not fit the compact record")
    return out


def make_events(x, y, t_us, sign=None):
    ev = np.zeros(len(x), dtype=EVENT_DTYPE)
    ev["x"] = x
    ev["y"] = y
    ev["t_us"] = t_us
    ev["sign"] = 1 if sign is None else sign
    return ev


class Graph:
    """A recorded step (ebo_graph): launch(times) replays it back to back on the context's stream."""

    def __init__(self, ctx, handle):
        self._ctx, self._g = ctx, handle

    def launch(self, times=1):
        self._ctx._check(lib().ebo_graph_launch(self._ctx._h, self._g, int(times)))

    def close(self):
        if self._g:
            lib().ebo_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Owns one ebo_ctx (one device, one stream).  Mirrors include/ebo.h 1:1."""

    def __init__(self, params=None, **kw):
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params(**kw)
        rc = lib().ebo_create(C.byref(self.params), C.byref(self._h))
        if rc:
            raise EboError(rc, lib().ebo_last_error(None).decode())
        npx, npy = C.c_int(), C.c_int()
        self._check(lib().ebo_grid(self._h, C.byref(npx), C.byref(npy)))
        self.npx, self.npy = npx.value, npy.value
        self.P = self.npx * self.npy
        self.n_windows = 0

    def _check(self, rc):
        if rc:
            raise EboError(rc, lib().ebo_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().ebo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- setup ---------------------------------------------------------------
    def set_stream(self, hip_stream):
        self._check(lib().ebo_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        self._check(lib().ebo_synchronize(self._h))

    def record(self, fn):
        """Record the asynchronous *_device calls fn() makes into a HIP graph (ebo_graph_begin / _end).
        Run fn() once before (work tables are allocated on first use).  -> Graph"""
        self._check(lib().ebo_graph_begin(self._h))
        try:
            fn()
        finally:
            g = C.c_void_p()
            rc = lib().ebo_graph_end(self._h, C.byref(g))
        self._check(rc)
        return Graph(self, g)

    def patch_rect(self, px, py):
        v = [C.c_int() for _ in range(4)]
        self._check(lib().ebo_patch_rect(self._h, int(px), int(py), *[C.byref(a) for a in v]))
        return tuple(a.value for a in v)

    def set_window(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self._check(lib().ebo_set_window(self._h, _vp(ev), C.c_size_t(len(ev))))
        self.n_windows = 1
        self._custom = None

    def set_windows(self, ev, offsets):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        self._check(lib().ebo_set_windows(self._h, _vp(ev), _vp(offsets), int(n)))
        self.n_windows = n
        self._custom = None

    def set_windows_device(self, d_events, offsets):
        """Raw ebo_event records already in device memory (pointer as int); offsets: host."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        self._check(lib().ebo_set_windows_device(self._h, C.c_void_p(int(d_events)), _vp(offsets), int(n)))
        self.n_windows = n
        self._custom = None

    def set_windows8(self, ev8, t_base, offsets, device=False):
        """Compact 8-byte records (EVENT8_DTYPE array or, with device=True, a device pointer)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        t_base = np.ascontiguousarray(t_base, dtype=np.int64)
        n = len(offsets) - 1
        assert len(t_base) == n
        if device:
            self._check(lib().ebo_set_windows8_device(self._h, C.c_void_p(int(ev8)), _vp(t_base), _vp(offsets), int(n)))
        else:
            if isinstance(ev8, np.ndarray):
                ev8 = np.ascontiguousarray(ev8, dtype=EVENT8_DTYPE)
                ptr = _vp(ev8)
            else:
                ptr = C.c_void_p(int(ev8))  # e.g. the address of page-locked memory
            self._check(lib().ebo_set_windows8(self._h, ptr, _vp(t_base), _vp(offsets), int(n)))
        self.n_windows = n
        self._custom = None

    def set_patches(self, ev, offsets, rects):
        """Arbitrary patches (contrastFunctor instances): rects [n][4] = (x, y, w, h)."""
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        rects = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
        self._check(lib().ebo_set_patches(self._h, _vp(ev), _vp(offsets), _vp(rects), len(rects)))
        self.n_windows = 1
        self._custom = [tuple(int(v) for v in r) for r in rects]

    @property
    def cur_patches(self):
        """Flows per window: grid patches, or the number given to set_patches."""
        return len(self._custom) if getattr(self, "_custom", None) else self.P

    def window_info(self, w=0):
        t, n = C.c_int64(), C.c_uint64()
        self._check(lib().ebo_window_info(self._h, int(w), C.byref(t), C.byref(n)))
        return t.value, n.value

    def patch_info(self, p, w=0):
        n, a, t = C.c_int32(), C.c_int32(), C.c_int64()
        self._check(lib().ebo_patch_info(self._h, int(w), int(p), C.byref(n), C.byref(a), C.byref(t)))
        return n.value, bool(a.value), t.value

    # -- objective -----------------------------------------------------------
    def eval(self, flows, want_jac=True):
        P = self.cur_patches
        flows = np.ascontiguousarray(flows, dtype=np.float64).reshape(self.n_windows, P, 2)
        r = np.zeros((self.n_windows, P))
        J = np.zeros((self.n_windows, P, 2)) if want_jac else None
        self._check(lib().ebo_eval(self._h, _dp(flows), _dp(r), _dp(J) if want_jac else None))
        return r, J

    def eval_device(self, d_flows, want_jac, d_out):
        self._check(lib().ebo_eval_device(
            self._h, C.c_void_p(int(d_flows)), int(want_jac), C.c_void_p(int(d_out))))

    def contrast_image(self, patch, flow, channels=3, window=0):
        if getattr(self, "_custom", None):
            x, y, w, h = self._custom[patch]
        else:
            x, y, w, h = self.patch_rect(patch % self.npx, patch // self.npx)
        flow = np.ascontiguousarray(flow, dtype=np.float64)
        img = np.zeros((channels, 3 * h, 3 * w))
        self._check(lib().ebo_contrast_image(
            self._h, int(window), int(patch), _dp(flow), int(channels), _dp(img)))
        return img

    # -- solve ---------------------------------------------------------------
    def solve(self, opts=None, **kw):
        opts = opts if opts is not None else default_solver(**kw)
        flows = np.zeros((self.n_windows, self.cur_patches, 2))
        summ = (Summary * self.n_windows)()
        self._check(lib().ebo_solve(self._h, C.byref(opts), _dp(flows), summ))
        return flows, list(summ)

    def solve_device(self, opts, d_flows_out, d_stats=0):
        self._check(lib().ebo_solve_device(
            self._h, C.byref(opts), C.c_void_p(int(d_flows_out)),
            C.c_void_p(int(d_stats)) if d_stats else None))

    # -- count images --------------------------------------------------------
    def count_image(self, mode, aux=None):
        img = np.zeros((self.n_windows, self.params.image_h, self.params.image_w))
        a = None
        if mode == COUNT_WARPED:
            aux = np.ascontiguousarray(aux, dtype=np.float64).reshape(self.n_windows, self.P, 2)
            a = _vp(aux)
        elif mode == COUNT_FIELD and aux is not None:
            aux = np.ascontiguousarray(aux, dtype=np.float32).reshape(
                self.n_windows, self.params.image_h, self.params.image_w, 2)
            a = _vp(aux)
        self._check(lib().ebo_count_image(self._h, int(mode), a, _dp(img)))
        return img

    def count_image_shard(self, n_windows, t_ref_us, flows_grid):
        """Partial final image (this context's events only) of n_windows windows whose patches are
        sharded: flows_grid [n_windows][P][2] = flows of ALL grid patches, t_ref_us [n_windows] = the
        windows' reference times.  Needs set_patches."""
        t_ref = np.ascontiguousarray(t_ref_us, dtype=np.int64).reshape(n_windows)
        flows = np.ascontiguousarray(flows_grid, dtype=np.float64).reshape(n_windows, self.P, 2)
        img = np.zeros((n_windows, self.params.image_h, self.params.image_w))
        self._check(lib().ebo_count_image_shard(self._h, int(n_windows), _vp(t_ref), _dp(flows), _dp(img)))
        return img

    def count_image_shard_device(self, n_windows, t_ref_us, d_flows_grid, d_image):
        t_ref = np.ascontiguousarray(t_ref_us, dtype=np.int64).reshape(n_windows)
        self._check(lib().ebo_count_image_shard_device(self._h, int(n_windows), _vp(t_ref),
                                                       C.c_void_p(int(d_flows_grid)), C.c_void_p(int(d_image))))

    def edge_work_stats(self, d_flows, want_jac=True):
        """diagnostic: one edge-loss evaluation that counts its work -> dict of totals over the units"""
        out = (C.c_uint64 * 6)()
        self._check(lib().ebo_edge_work_stats(self._h, C.c_void_p(int(d_flows)), 1 if want_jac else 0, out))
        keys = ("units", "events", "box_pixels", "eigen_pixels", "nms_windows", "argmax_entries")
        return dict(zip(keys, [int(v) for v in out]))

    def lds_rates(self):
        """diagnostic: (G atomics/s, G reads/s) of 64-bit LDS operations at random addresses, measured now"""
        out = (C.c_double * 2)()
        self._check(lib().ebo_lds_rates(self._h, out))
        return float(out[0]), float(out[1])

    def stream_yardstick_device(self, d_image):
        """diagnostic: the bytes of a count-image launch with no work; returns the bytes moved"""
        n = C.c_uint64()
        self._check(lib().ebo_stream_yardstick_device(self._h, C.c_void_p(int(d_image)), C.byref(n)))
        return n.value

    def count_image_device(self, mode, d_aux, d_image):
        self._check(lib().ebo_count_image_device(
            self._h, int(mode), C.c_void_p(int(d_aux)) if d_aux else None,
            C.c_void_p(int(d_image))))

    def compensate_events_contrast(self, ev, opts=None, want_image=True):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        opts = opts if opts is not None else default_solver()
        flows = np.zeros((self.P, 2))
        img = np.zeros((self.params.image_h, self.params.image_w)) if want_image else None
        s = Summary()
        self._check(lib().ebo_compensate_events_contrast(
            self._h, _vp(ev), C.c_size_t(len(ev)), C.byref(opts), _dp(flows),
            _dp(img) if want_image else None, C.byref(s)))
        self.n_windows = 1
        self._custom = None
        return flows, img, s

    def init_motion_field(self, timestamp, trajectories, use_average=True):
        """FeatureDetector::initMotionField. trajectories: list of [(x, y, t_us), ...] per patch.
        Returns (field float32 [H][W][2], fixed points [n][2])."""
        offs = np.zeros(len(trajectories) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(t) for t in trajectories])
        flat = [s for t in trajectories for s in t]
        xy = np.ascontiguousarray([[s[0], s[1]] for s in flat], dtype=np.float64).reshape(-1, 2)
        tt = np.ascontiguousarray([int(s[2]) for s in flat], dtype=np.int64)
        field = np.zeros((self.params.image_h, self.params.image_w, 2), dtype=np.float32)
        nfix = C.c_int32()
        fixed = np.zeros((max(len(trajectories), 1), 2), dtype=np.int32)
        self._check(lib().ebo_init_motion_field(
            self._h, C.c_int64(int(timestamp)), int(bool(use_average)), len(trajectories), _vp(offs),
            _vp(xy), _vp(tt), _vp(field), C.byref(nfix), _vp(fixed)))
        return field, fixed[: nfix.value].copy()

    def interpolate_motion_field(self, use_l1=False, opts=None):
        """FeatureDetector::interpolateMotionField on the field of the last init_motion_field.
        Returns (field float32 [H][W][2], Summary, total CG iterations)."""
        field = np.zeros((self.params.image_h, self.params.image_w, 2), dtype=np.float32)
        s = Summary()
        cg = C.c_int32()
        self._check(lib().ebo_interpolate_motion_field(
            self._h, int(bool(use_l1)), C.byref(opts) if opts is not None else None, _vp(field),
            C.byref(s), C.byref(cg)))
        return field, s, cg.value

    # -- per-feature tracker objective (Optimizer / OptimizerCostFunctor) ------
    def optimizer_set_grad(self, grad_x, grad_y):
        gx = np.ascontiguousarray(grad_x, dtype=np.float64)
        gy = np.ascontiguousarray(grad_y, dtype=np.float64)
        assert gx.shape == gy.shape == (self.params.image_h, self.params.image_w)
        self._check(lib().ebo_optimizer_set_grad(self._h, _dp(gx), _dp(gy)))

    @staticmethod
    def _opt_inputs(rects, nablas, poses, flow_dirs):
        rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        nabla = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in nablas])
                                     if len(nablas) else np.zeros(0))
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4).copy()
        flow_dirs = np.ascontiguousarray(flow_dirs, dtype=np.float64).reshape(-1).copy()
        sizes = [int(r[2]) * int(r[3]) for r in rects]
        assert nabla.size == sum(sizes)
        return rects, nabla, poses, flow_dirs, sizes

    def optimizer_eval(self, rects, nablas, poses, flow_dirs, want_jac=True):
        """OptimizerCostFunctor for n patches.  Returns lists (residuals, jac_pose, jac_flow)."""
        rects, nabla, poses, flow_dirs, sizes = self._opt_inputs(rects, nablas, poses, flow_dirs)
        total = sum(sizes)
        res = np.zeros(total)
        jp = np.zeros((total, 4)) if want_jac else None
        jf = np.zeros(total) if want_jac else None
        self._check(lib().ebo_optimizer_eval(
            self._h, len(rects), _dp(rects), _dp(nabla), _dp(poses), _dp(flow_dirs), _dp(res),
            _dp(jp) if want_jac else None, _dp(jf) if want_jac else None))
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        cut = lambda a: [a[offs[i]:offs[i + 1]] for i in range(len(sizes))]
        return cut(res), (cut(jp) if want_jac else None), (cut(jf) if want_jac else None)

    def optimizer_solve(self, rects, nablas, poses, flow_dirs, normalize=False, huber=0.3, opts=None):
        """Optimizer::optimize's ceres::Solve for n patches.  Returns (poses, flow_dirs, summaries)."""
        rects, nabla, poses, flow_dirs, sizes = self._opt_inputs(rects, nablas, poses, flow_dirs)
        n = len(rects)
        sums = (Summary * max(n, 1))()
        self._check(lib().ebo_optimizer_solve(
            self._h, n, _dp(rects), _dp(nabla), int(bool(normalize)), C.c_double(huber),
            C.byref(opts) if opts is not None else None, _dp(poses), _dp(flow_dirs), sums))
        return poses, flow_dirs, list(sums)[:n]

    def optimizer_cost_map(self, rects, nablas, poses, flow_dirs, map_w=11, map_h=11, normalize=False):
        """Optimizer::drawCostMap for n patches: -> [n][map_h][map_w] (the L2 norm of the functor's residual image
        at the map's translation offsets around each pose)."""
        rects, nabla, poses, flow_dirs, sizes = self._opt_inputs(rects, nablas, poses, flow_dirs)
        n = len(rects)
        out = np.zeros((max(n, 1), map_h, map_w))
        self._check(lib().ebo_optimizer_cost_map(
            self._h, n, _dp(rects), _dp(nabla), int(bool(normalize)), _dp(poses), _dp(flow_dirs), int(map_w), int(map_h),
            _dp(out)))
        return out[:n]

    def estimate_num_events(self, rects, poses, flow_dirs):
        """FeatureDetector::updateNumOfEvents' event-count estimate for n tracked patches."""
        rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4)
        flow_dirs = np.ascontiguousarray(flow_dirs, dtype=np.float64).reshape(-1)
        n = len(rects)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self._check(lib().ebo_estimate_num_events(self._h, n, _dp(rects), _dp(poses), _dp(flow_dirs), _vp(out)))
        return out[:n]

    def patch_warp_image(self, rects, poses, flow_dirs):
        """Patch::warpImage for n tracked patches: -> list of predictedNabla arrays [h][w] (None where the
        reference returns early because the rect touches the image border)."""
        rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4)
        flow_dirs = np.ascontiguousarray(flow_dirs, dtype=np.float64).reshape(-1)
        n = len(rects)
        shapes = [(int(np.rint(r[3])), int(np.rint(r[2]))) for r in rects]
        sizes = [h * w for h, w in shapes]
        noff = np.zeros(max(n, 1), dtype=np.uint64)
        noff[1:n] = np.cumsum(sizes[:-1])
        out = np.zeros(max(int(sum(sizes)), 1))
        upd = np.zeros(max(n, 1), dtype=np.int32)
        self._check(lib().ebo_patch_warp_image(self._h, n, _dp(rects), _dp(poses), _dp(flow_dirs), _vp(noff), _dp(out),
                                               _vp(upd)))
        return [out[int(noff[i]):int(noff[i]) + sizes[i]].reshape(shapes[i]).copy() if upd[i] else None for i in range(n)]

    # -- tracked-feature patches (Patch::integrate*) --------------------------
    def patch_integrate(self, ev, offsets, rects):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        n = len(rects)
        sizes = [int(r[3]) * int(r[2]) for r in rects]
        noff = np.zeros(n, dtype=np.uint64)
        noff[1:] = np.cumsum(sizes[:-1])
        nabla = np.zeros(int(sum(sizes)))
        cur = np.zeros(n, dtype=np.int64)
        last = np.zeros(n, dtype=np.int64)
        self._check(lib().ebo_patch_integrate(
            self._h, _vp(ev), _vp(offsets), n, _dp(rects), _vp(noff), _dp(nabla), _vp(cur), _vp(last)))
        imgs = [nabla[int(noff[i]):int(noff[i]) + sizes[i]].reshape(int(rects[i][3]), int(rects[i][2]))
                for i in range(n)]
        return imgs, cur, last

    def patch_integrate_mc(self, ev, offsets, rects, traj, mid_time, init=None):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        traj = np.ascontiguousarray(traj, dtype=np.float64).reshape(-1, 6)
        mid_time = np.ascontiguousarray(mid_time, dtype=np.int64)
        n = len(rects)
        sizes = [int(r[3]) * int(r[2]) for r in rects]
        noff = np.zeros(n, dtype=np.uint64)
        noff[1:] = np.cumsum(sizes[:-1])
        nabla = np.zeros(int(sum(sizes))) if init is None else np.ascontiguousarray(init, dtype=np.float64).copy()
        upd = np.zeros(n, dtype=np.int32)
        self._check(lib().ebo_patch_integrate_mc(
            self._h, _vp(ev), _vp(offsets), n, _dp(rects), _dp(traj), _vp(mid_time), _vp(noff),
            _dp(nabla), _vp(upd)))
        imgs = [nabla[int(noff[i]):int(noff[i]) + sizes[i]].reshape(int(rects[i][3]), int(rects[i][2]))
                for i in range(n)]
        return imgs, upd

    # -- event -> tracked-patch routing (FeatureDetector::updatePatches) --------
    def route_set_events(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self._check(lib().ebo_route_set_events(self._h, _vp(ev), C.c_size_t(len(ev))))

    def route_events(self, rects, start, max_take, cap):
        """-> (list of index arrays per patch, next index per patch)."""
        rects = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        n = len(rects)
        start = np.ascontiguousarray(start, dtype=np.uint32)
        max_take = np.ascontiguousarray(max_take, dtype=np.uint32)
        idx = np.zeros((n, max(int(cap), 1)), dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint32)
        nxt = np.zeros(n, dtype=np.uint32)
        self._check(lib().ebo_route_events(self._h, n, _dp(rects), _vp(start), _vp(max_take), C.c_uint32(int(cap)),
                                           _vp(idx), _vp(cnt), _vp(nxt)))
        return [idx[i, :cnt[i]].copy() for i in range(n)], nxt

    # -- RCCL exchange (no framework) -----------------------------------------
    def comm_init(self, comm_id, rank, nranks):
        buf = (C.c_char * 128).from_buffer_copy(bytes(comm_id))
        self._check(lib().ebo_comm_init(self._h, C.byref(buf), int(rank), int(nranks)))
        self._nranks = int(nranks)

    def allgather_device(self, d_send, d_recv, count_per_rank):
        self._check(lib().ebo_allgather_device(
            self._h, C.c_void_p(int(d_send)), C.c_void_p(int(d_recv)), C.c_size_t(int(count_per_rank))))

    def comm_size(self):
        """(rank, nranks) of the context's communicator; (0, 1) without one."""
        r, n = C.c_int(), C.c_int()
        self._check(lib().ebo_comm_size(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def reduce_sum_device(self, d_send, d_recv, count, root=-1):
        """Sum of `count` doubles over the ranks into d_recv on `root` (every rank when root < 0)."""
        self._check(lib().ebo_reduce_sum_device(
            self._h, C.c_void_p(int(d_send)), C.c_void_p(int(d_recv)) if d_recv else None, C.c_size_t(int(count)),
            int(root)))

    def allgather_tracks(self, local):
        """Every rank's track records on every rank (rank order): -> (records, counts per rank).  ONE counts
        collective + ONE gather per call: the result buffer is kept between calls and grows by a rule that
        depends only on the gathered total (4096 records, then the next power of two), so every rank finds it
        too small -- and repeats the exchange -- in the same call."""
        local = np.ascontiguousarray(local, dtype=TRACK_DTYPE)
        nr = C.c_size_t()
        counts = np.zeros(self.comm_size()[1], dtype=np.uint64)
        buf = getattr(self, "_track_buf", None)
        if buf is None:
            buf = self._track_buf = np.zeros(4096, dtype=TRACK_DTYPE)
        rc = lib().ebo_allgather_tracks(self._h, _vp(local), C.c_size_t(len(local)), _vp(buf), C.c_size_t(len(buf)),
                                        C.byref(nr), _vp(counts))
        if rc == ERR_ARG and nr.value > len(buf):  # the same on every rank: all of them repeat
            buf = self._track_buf = np.zeros(1 << int(nr.value - 1).bit_length(), dtype=TRACK_DTYPE)
            rc = lib().ebo_allgather_tracks(self._h, _vp(local), C.c_size_t(len(local)), _vp(buf), C.c_size_t(len(buf)),
                                            C.byref(nr), _vp(counts))
        self._check(rc)
        return buf[:nr.value].copy(), counts.astype(np.int64)

    # -- band-limited final image of row-sharded windows -------------------------
    def band_plan(self, row_bounds, rank, halo):
        """-> Band for `rank` of the partition row_bounds [nranks + 1] (image rows); EboError(ERR_UNSUPPORTED) when a
        halo would not fit inside a neighbour's rows."""
        return band_plan(self.params.image_h, row_bounds, rank, halo)

    def count_image_band_device(self, n_windows, t_ref_us, d_flows_grid, band, d_top, d_own, d_bottom, d_escaped):
        t_ref = np.ascontiguousarray(t_ref_us, dtype=np.int64).reshape(n_windows)
        self._check(lib().ebo_count_image_band_device(
            self._h, int(n_windows), _vp(t_ref), C.c_void_p(int(d_flows_grid)), C.byref(band),
            C.c_void_p(int(d_top)) if d_top else None, C.c_void_p(int(d_own)) if d_own else None,
            C.c_void_p(int(d_bottom)) if d_bottom else None, C.c_void_p(int(d_escaped))))

    def band_exchange_device(self, n_windows, band, d_top, d_bottom, d_from_above, d_from_below, d_escaped):
        p = lambda v: C.c_void_p(int(v)) if v else None
        self._check(lib().ebo_band_exchange_device(self._h, int(n_windows), C.byref(band), p(d_top), p(d_bottom),
                                                   p(d_from_above), p(d_from_below), p(d_escaped)))

    def band_finish_device(self, n_windows, band, d_own, d_from_above, d_from_below, d_image_own):
        p = lambda v: C.c_void_p(int(v)) if v else None
        self._check(lib().ebo_band_finish_device(self._h, int(n_windows), C.byref(band), p(d_own), p(d_from_above),
                                                 p(d_from_below), p(d_image_own)))

    def band_gather_device(self, n_windows, row_bounds, d_image_own, root, d_full):
        rb = np.ascontiguousarray(row_bounds, dtype=np.int32)
        self._check(lib().ebo_band_gather_device(self._h, int(n_windows), _vp(rb), C.c_void_p(int(d_image_own)) if d_image_own else None,
                                                 int(root), C.c_void_p(int(d_full)) if d_full else None))

    def comm_destroy(self):
        self._check(lib().ebo_comm_destroy(self._h))

    # -- timing --------------------------------------------------------------
    def timer_begin(self):
        self._check(lib().ebo_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float()
        self._check(lib().ebo_timer_end(self._h, C.byref(ms)))
        return ms.value
