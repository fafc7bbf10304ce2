#!/usr/bin/env python3
"""bench.py — Mevents/s warped+scored on MI355X (BASELINE.json's metric), at every N.

    python bench.py --gpus N --steps K --warmup W

One event-evaluation = one event warped by its patch's candidate flow, splatted and scored once
(SURVEY §8(d)).  Inputs (packed events, unit tables, flows) are resident in HBM before the timed
region.  Three workloads; `--workload auto` (default) picks by N:

  eval      N = 1 default.  A step = ONE batched value+Jacobian evaluation of the variance-contrast
            objective (EBO_GRAD_JET) over `--windows` independent windows of BASELINE configs[2]
            (C3: 346x260, 256 patches, 200 k events/window — the largest single-GPU configuration).
            With N > 1 every rank evaluates its own windows, no data-path collective.
  c4        N > 1 default = BASELINE configs[3]: 1280x720 windows of 1024 patches / 1 M events whose
            PATCH ROWS are sharded over the N GPUs (ebo_shard_range + ebo_set_patches: 32 grid
            rows -> 32/N rows = 1024/N patches per window per GPU).  A step = the device-resident
            per-patch solve of the rank's shard (ebo_solve_device, one launch) + ONE RCCL all-gather of
            the solved flows (16 B per patch), after which every rank holds every window's flows, + the
            last third of compensateEventsContrast (feature_detector.cpp:433-463): every rank counts ITS
            events warped by the gathered flows at the window's reference time (ebo_count_image_shard)
            and ONE reduce sums the integer-valued partial images on rank 0 (`--no-c4-image` leaves
            the step at solve + all-gather, as round 2 measured it).
            `--c4-windows` windows PER GPU are in flight (N x that many windows in the batch), so
            the per-GPU work is fixed as N grows: weak scaling.  value = event-evaluations of the
            solves (events x objective evaluations, from the solver's own statistics) per second.
  replicas  BASELINE configs[4]: N independent 346x260 sequences, one per GPU.  A step = the rank's
            batched evaluation (as `eval`) + one solve of the rank's tracked feature patches
            (ebo_optimizer_solve) whose new trajectory points join the rank's tracks + ONE gather of
            all ranks' variable-length (id, t, x, y) track lists (counts + max-padded all-gather).

N > 1 without a launcher: `python bench.py --gpus N` spawns `python -m torch.distributed.run
--nproc-per-node N` on itself BEFORE anything touches the GPU and relays the children's output;
under torchrun (WORLD_SIZE set) it just runs as rank RANK.  EBO_BENCH_REHEARSE=1 rehearses the
N > 1 code path on a box with fewer GPUs than ranks (all ranks share the visible GPUs, collectives
over gloo) and says so in the JSON line — never a scaling measurement.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

T_START = time.time() - 2.0  # files of the id hand-over older than this run are somebody else's
# how long a rank waits for another rank's file of the id hand-over / the vote (rank 0 comes from its CPU baseline,
# 10-40 s later than the others): beyond it the library's communicator counts as failed on that rank
COMM_TIMEOUT_S = float(os.environ.get("EBO_BENCH_COMM_TIMEOUT_S", "120"))
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X spec: f64 vector FMA, 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz
# Useful f64 operations of one edge-loss evaluation per item of work (DESIGN.md 4.5 derives them from
# ebo_edge.inc; an FMA counts 2, an exp as its 22-operation polynomial, comparisons and address
# arithmetic count nothing): scatter per event, conversion per box pixel, separable tensor filter +
# eigenvalue per eigenvalue-region pixel (8-row runs: 14 / 8 row steps of 77 + 53), the eigenvector
# direction per pixel of a Jacobian evaluation, reverse sweep per argmax entry, gather per event.
EDGE_FLOPS = {"scatter_per_event": 290, "convert_per_box_pixel": 3, "eigen_per_pixel": 188, "direction_per_pixel": 14,
              "reverse_per_entry": 860, "gather_per_event": 420}


def edge_flops(stats, want_jac):
    f = (stats["events"] * EDGE_FLOPS["scatter_per_event"] + stats["box_pixels"] * EDGE_FLOPS["convert_per_box_pixel"]
         + stats["eigen_pixels"] * EDGE_FLOPS["eigen_per_pixel"])
    if want_jac:
        f += (stats["eigen_pixels"] * EDGE_FLOPS["direction_per_pixel"] + stats["argmax_entries"] * EDGE_FLOPS["reverse_per_entry"]
              + stats["events"] * EDGE_FLOPS["gather_per_event"])
    return float(f)
BYTES_PER_EVENT_EVAL = 8  # packed {x:15, pol:1, y:15, dt:32}; SURVEY §8(d)


# --------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU restatement of the reference path; the reference itself cannot be
# built in this image) timed on this box's host cores, single thread like the reference (F1).
# --------------------------------------------------------------------------------------------
def cpu_baseline(synth, cfg_idx, seconds_budget):
    import orc
    cfg = synth.CONFIGS[cfg_idx]
    ev, gt = synth.make_window(cfg_idx)
    prm = orc.default_params(image_w=cfg["image"][0], image_h=cfg["image"][1],
                             patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=1, tv_weight=0.0)
    flows = gt * 0.5
    sec, n = orc.window_eval_timed(ev, prm, flows, True, 1)  # warm-up + calibration
    reps = max(1, min(2000, int(seconds_budget / max(sec, 1e-6))))
    sec, n = orc.window_eval_timed(ev, prm, flows, True, reps)
    base = {
        "value": n / sec / 1e6, "unit": "Mevents/s", "cores": 1, "kind": "port",
        "flags": "-O2 -ffp-contract=off (oracle/liboracle.so, the tests' checker)",
        "sample": "%d value+Jacobian evaluations of 1 window of %s (%d event-evaluations, %.1f s) "
                  "by oracle/liboracle.so, single thread like the reference" % (reps, cfg["name"], n, sec),
    }
    # The same sample on the reference's own optimisation flags (CMakeLists.txt:43,57-60: -O3 -DNDEBUG -march=native),
    # built HERE, on the machine that runs it, into a scratch directory; -ffp-contract=off kept (oracle/Makefile).
    try:
        import ctypes as C
        import subprocess
        import tempfile
        scratch = tempfile.mkdtemp(prefix="ebo_oracle_")
        so = os.path.join(scratch, "liboracle_ref_flags.so")
        subprocess.check_call(["make", "-s", "-B", "-C", orc.ORACLE_DIR, "liboracle_ref_flags.so", "OUT_REF=" + so],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        fast = C.CDLL(so)
        evc = np.ascontiguousarray(ev, dtype=orc.EVENT_DTYPE)
        fl = np.ascontiguousarray(flows, dtype=np.float64)
        s2, c2 = C.c_double(), C.c_uint64()
        for r in (1, reps):
            rc = fast.orc_window_eval_timed(evc.ctypes.data_as(C.c_void_p), C.c_size_t(len(evc)), C.byref(prm),
                                            fl.ctypes.data_as(C.POINTER(C.c_double)), 1, int(r), C.byref(s2), C.byref(c2))
            assert rc == 0
        base["value_ref_flags"] = c2.value / s2.value / 1e6
        base["flags_ref"] = "-O3 -DNDEBUG -march=native -ffp-contract=off (the reference's CMakeLists.txt:43,57-60), built on this host"
    except Exception as exc:  # a reported extra: the -O2 figure above stands
        base["value_ref_flags"] = None
        base["flags_ref"] = "not built: %r" % (exc,)
    legs = {}
    # (ii) a full per-patch solve of one window (the reference-default configuration: 15 k events,
    # 108 patches of 20x20, the solve the `c4` workload runs per shard), event-evaluations/s
    try:
        ev0, _ = synth.make_window(0)
        prm0 = orc.default_params(loss=1, tv_weight=0.0)
        opts = orc.default_solver(mode=1)
        t0 = time.perf_counter()
        _, _, summ = orc.compensate_events_contrast(ev0, prm0, opts, want_image=False)
        dt = time.perf_counter() - t0
        n_act, p_act = 0, 0
        npx, npy = orc.grid(prm0)
        for p in range(npx * npy):
            x, y, w, h = orc.patch_rect(prm0, p % npx, p // npx)
            k = int(((ev0["x"] >= x) & (ev0["x"] < x + w) & (ev0["y"] >= y) & (ev0["y"] < y + h)).sum())
            if k > prm0.min_events:
                n_act += k
                p_act += 1
        evals = int(summ.num_evals_cost + summ.num_evals_jac)  # summed over the patches
        legs["solve_independent"] = {
            "value": n_act * (evals / max(p_act, 1)) / dt / 1e6, "unit": "Mevents/s", "cores": 1,
            "sample": "1 per-patch solve of 1 window of %s (%d scored events in %d patches, %d evaluations over all "
                      "patches, %.2f s; events x mean evaluations per patch)"
                      % (synth.CONFIGS[0]["name"], n_act, p_act, evals, dt)}
    except Exception as exc:  # a reported extra
        legs["solve_independent"] = {"error": repr(exc)}
    # (iii) the integer count images alone (feature_detector.cpp:433-463 warped, :466-482 un-warped)
    try:
        reps_c = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < min(3.0, seconds_budget):
            orc.final_count_image(ev, prm, flows)
            reps_c += 1
        dt = time.perf_counter() - t0
        legs["count_image_warped"] = {"value": len(ev) * reps_c / dt / 1e6, "unit": "Mevents/s", "cores": 1,
                                      "sample": "%d warped count images of 1 window of %s" % (reps_c, cfg["name"])}
        reps_c = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < min(2.0, seconds_budget):
            orc.integrate_events(ev, cfg["image"][0], cfg["image"][1])
            reps_c += 1
        dt = time.perf_counter() - t0
        legs["count_image_integrated"] = {"value": len(ev) * reps_c / dt / 1e6, "unit": "Mevents/s", "cores": 1,
                                          "sample": "%d un-warped count images of 1 window of %s" % (reps_c, cfg["name"])}
    except Exception as exc:
        legs["count_image"] = {"error": repr(exc)}
    base["legs"] = legs
    return base


def _cpu_worker(args):
    """One host core: value+Jacobian evaluations of its own synthetic window by the oracle."""
    cfg_idx, window, seconds = args
    import orc
    synth = importlib.import_module("event-based-odomety_amd.synth")
    cfg = synth.CONFIGS[cfg_idx]
    ev, gt = synth.make_window(cfg_idx, window=window)
    prm = orc.default_params(image_w=cfg["image"][0], image_h=cfg["image"][1],
                             patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=1, tv_weight=0.0)
    flows = gt * 0.5
    sec, n = orc.window_eval_timed(ev, prm, flows, True, 1)
    reps = max(1, min(2000, int(seconds / max(sec, 1e-6))))
    sec, n = orc.window_eval_timed(ev, prm, flows, True, reps)
    return n, sec


def cpu_baseline_all_cores(cfg_idx, seconds):
    """The same oracle on every host core at once (independent windows are independent problems;
    SURVEY §8(d): a parallel CPU variant with the core count stated).  Runs in forked workers
    BEFORE this process touches the GPU."""
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(cfg_idx, 1000 + k, seconds) for k in range(cores)])
    rate = sum(n / sec for n, sec in res) / 1e6
    return {"value": rate, "unit": "Mevents/s", "cores": cores, "kind": "port",
            "sample": "%d processes x ~%.0f s of value+Jacobian evaluations, one window each, by oracle/liboracle.so"
                      % (cores, seconds)}


# --------------------------------------------------------------------------------------------
# launcher
# --------------------------------------------------------------------------------------------
LDS_PEAK_GOPS = {"ds_add_u64_random": 4335.0, "ds_read_b64_random": 8070.0}  # profiles/r04_lds_rates.txt, "7x7 taps, any base" (fallback only)


def lds_roofline(event_evals, kern_ms, rates=None):
    """The LDS-side bound of one value+Jacobian evaluation launch: time the scatter's atomics and the
    gather's reads would take at the LDS rates measured IN THIS RUN on this GPU (ebo_lds_rates: the kernel's own
    access shape -- a random base slot per lane, then the 49 taps of a 7 x 7 footprint at immediate offsets, 16 waves per
    CU; tools/microbench/lds_atomics.hip "7x7 taps, any base"), against the measured launch.  Since round 4's last
    kernels the probe has that shape; before, it issued ONE random operation per loop trip and read 15-25 % lower,
    i.e. earlier rounds' lds.frac figures are that much too high against this yardstick."""
    peaks = dict(LDS_PEAK_GOPS) if rates is None else {"ds_add_u64_random": rates[0], "ds_read_b64_random": rates[1]}
    min_ms = (49.0 * event_evals / (peaks["ds_add_u64_random"] * 1e9)
              + 49.0 * event_evals / (peaks["ds_read_b64_random"] * 1e9)) * 1e3
    return {"ops_per_event": {"ds_add_u64": 49, "ds_read_b64": 49}, "peak_Gops": peaks,
            "min_ms": min_ms, "frac": min_ms / kern_ms,
            "source": "measured in this run (ebo_lds_rates)" if rates is not None
            else "event-based-odomety_amd/tools/microbench/lds_atomics.hip (profiles/r04_lds_rates.txt)",
            "shape": "random base per lane + 49 taps at immediate offsets (the kernel's own), 16 waves per CU"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_children(cmd, env, deadline_s):
    """Run `cmd` as a child in its OWN session and wait at most deadline_s for it: beyond that its process group --
    exactly the processes this call started -- is ended (SIGTERM, then SIGKILL) and the call returns 124 after a
    one-line diagnosis on stderr.  A multi-rank run that hangs on first contact (a communicator that never comes up,
    a rank that never reaches a collective) must end as a failed run with a reason, below the time the driver gives a
    bench, not as a silent time-limit kill."""
    import signal

    def child_setup():
        # the child leads its own session (so that its whole group can be ended), and must not outlive this process:
        # if the parent is killed outright the kernel sends the child SIGTERM (torch.distributed.run then ends its ranks)
        try:
            import ctypes
            ctypes.CDLL(None).prctl(1, int(signal.SIGTERM))  # PR_SET_PDEATHSIG
        except Exception:
            pass
    proc = subprocess.Popen(cmd, env=env, start_new_session=True, preexec_fn=child_setup)

    def forward(signum, _frame):  # a signal to the launcher ends the ranks too, then the launcher
        try:
            os.killpg(proc.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        try:
            proc.wait(timeout=15)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        sys.exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sg, forward)
        except (ValueError, OSError):  # not the main thread
            pass
    try:
        return proc.wait(timeout=deadline_s)
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench.py: the ranks did not finish within %.0f s (EBO_BENCH_DEADLINE_S) -- ending them; "
                         "no result line\n" % deadline_s)
        sys.stderr.flush()
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(proc.pid, sig)  # the session this call created: pgid == the child's pid
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start one rank per GPU as CHILD
    processes (torch.distributed.run) before this process has touched the GPU, relay their output,
    exit with their code.  Nothing is exec'ed and this process never initialises HIP.  The children get
    EBO_BENCH_DEADLINE_S seconds (default 540: below the driver's 600) before they are ended with a diagnosis."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return run_children(cmd, env, float(os.environ.get("EBO_BENCH_DEADLINE_S", "540")))


class TorchComm:
    """The collectives of one run over torch.distributed: RCCL ("nccl") on device tensors, or -- rehearsal only --
    gloo on host copies.  world == 1 without EBO_BENCH_FORCE_DIST: no process group at all."""

    def __init__(self, torch, rank, world, local, rehearse, force):
        self.torch, self.rank, self.world = torch, rank, world
        self.active = world > 1 or force
        self.backend = None
        if self.active:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            self.backend = "gloo" if rehearse else "nccl"
            if rehearse:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        self.transport = "none (one rank)" if not self.active else (
            "torch.distributed over %s" % ("RCCL" if self.backend == "nccl" else self.backend))

    def attach(self, ctx):
        pass

    @property
    def coll_device(self):
        return "cpu" if self.backend == "gloo" else "cuda"

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.active:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, value, op):
        if not self.active:
            return float(value)
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.coll_device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def allgather_rows(self, exchange, t_local, counts, out):
        """ONE all-gather of per-rank row blocks (max-padded when unequal) into `out`."""
        if not self.active:
            out.copy_(t_local)
            return out
        if self.backend == "gloo":
            res = exchange.allgather_rows(t_local.cpu(), counts)
            out.copy_(res)
            return out
        return exchange.allgather_rows(t_local, counts, out=out)

    def reduce_sum_to_rank0(self, t_dev, staging=None):
        """ONE reduce (sum) of a device tensor onto rank 0, in place there.  staging: an int32 device tensor of
        the same shape -- for integer-valued data (event counts, far below 2^31) the collective then moves half the
        bytes of the f64 image and is exact all the same."""
        if not self.active:
            return t_dev
        src = t_dev
        if staging is not None:
            staging.copy_(t_dev)  # f64 -> int32, exact on counts
            src = staging
        if self.backend == "gloo":
            h = src.cpu()
            self.dist.reduce(h, dst=0, op=self.dist.ReduceOp.SUM)
            if self.rank == 0:
                t_dev.copy_(h)
            return t_dev
        self.dist.reduce(src, dst=0, op=self.dist.ReduceOp.SUM)
        if staging is not None and self.rank == 0:
            t_dev.copy_(staging)
        return t_dev

    def halo_exchange(self, exchange, ctx, n_windows, band, B):
        """top -> rank - 1, bottom -> rank + 1, the neighbours' halos in, the flag = max over ranks"""
        if not self.active:
            return
        if self.backend == "gloo":
            host = {k: (B[k].cpu() if B[k] is not None else None) for k in ("top", "bottom", "from_above", "from_below")}
            flag = B["flag"].cpu()
            exchange.halo_exchange(host["top"], host["bottom"], host["from_above"], host["from_below"], flag)
            for k in ("from_above", "from_below"):
                if B[k] is not None:
                    B[k].copy_(host[k])
            B["flag"].copy_(flag)
            return
        exchange.halo_exchange(B["top"], B["bottom"], B["from_above"], B["from_below"], B["flag"])

    def allgather_tracks(self, exchange, ctx, local):
        if not self.active:
            return local, [len(local)]
        return exchange.allgather_tracks(local, device=self.coll_device)

    def close(self):
        if self.active:
            self.dist.destroy_process_group()


class EboComm:
    """The same collectives on the LIBRARY's own communicator (csrc/ebo_comm.cpp: librccl.so dlopened, no
    framework): rank 0 makes the id (ebo_comm_unique_id), hands its 128 bytes over through a file
    (exchange.handover_bytes), every rank calls ebo_comm_init on the workload's context; barrier and reductions are
    ebo_allgather_device of one double per rank, the flows go through ebo_allgather_device, the halos through
    ebo_band_exchange_device, the tracks through ebo_allgather_tracks."""

    def __init__(self, torch, ebo, exchange, rank, world):
        self.torch, self.ebo, self.exchange, self.rank, self.world = torch, ebo, exchange, rank, world
        self.active = True
        self.backend = "ebo"
        self.transport = "libebo_hip.so's own RCCL communicator (ebo_comm_init)"
        self.ctx = None
        self.prefix = os.path.join("/tmp", "ebo_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))

    def attach(self, ctx):
        cid, why = None, None
        if self.rank == 0:
            try:
                cid = self.ebo.comm_unique_id()
            except Exception as exc:  # librccl not loadable, ncclGetUniqueId failed: the other ranks must learn it NOW
                why = exc
        # (rank 0 arrives after its CPU baseline, seconds later than the others; a failed id is handed over as such)
        cid = self.exchange.handover_bytes(self.prefix, self.rank, self.world, cid, timeout=COMM_TIMEOUT_S, not_before=T_START)
        if why is not None:
            raise why
        ctx.comm_init(cid, self.rank, self.world)
        self.ctx = ctx
        self.d_one = self.torch.zeros(1, dtype=self.torch.float64, device="cuda")
        self.d_all = self.torch.zeros(self.world, dtype=self.torch.float64, device="cuda")
        self.d_pad = None

    def _gather_scalar(self, value):
        self.d_one.fill_(float(value))
        self.ctx.allgather_device(self.d_one.data_ptr(), self.d_all.data_ptr(), 1)
        self.torch.cuda.synchronize()
        return self.d_all.cpu().numpy()

    def barrier(self):
        self.torch.cuda.synchronize()
        self._gather_scalar(0.0)  # returns on a rank only after every rank has entered

    def reduce(self, value, op):
        v = self._gather_scalar(value)
        return float(v.max() if op == "MAX" else v.sum())

    def allgather_rows(self, exchange, t_local, counts, out):
        mx = max(counts)
        tail = int(np.prod(t_local.shape[1:])) if t_local.dim() > 1 else 1
        if min(counts) == mx:
            self.ctx.allgather_device(t_local.data_ptr(), out.data_ptr(), mx * tail)
            return out
        if self.d_pad is None or self.d_pad[0].shape[0] != mx:
            self.d_pad = (self.torch.zeros((mx,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device="cuda"),
                          self.torch.zeros((self.world * mx,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device="cuda"))
        pad, full = self.d_pad
        pad[: counts[self.rank]].copy_(t_local)
        self.ctx.allgather_device(pad.data_ptr(), full.data_ptr(), mx * tail)
        at = 0
        for q in range(self.world):
            out[at:at + counts[q]].copy_(full[q * mx:q * mx + counts[q]])
            at += counts[q]
        return out

    def reduce_sum_to_rank0(self, t_dev, staging=None):
        self.ctx.reduce_sum_device(t_dev.data_ptr(), t_dev.data_ptr(), t_dev.numel(), root=0)
        return t_dev

    def halo_exchange(self, exchange, ctx, n_windows, band, B):
        ptr = lambda t: t.data_ptr() if t is not None else 0
        ctx.band_exchange_device(n_windows, band, ptr(B["top"]), ptr(B["bottom"]), ptr(B["from_above"]), ptr(B["from_below"]),
                                 B["flag"].data_ptr())

    def allgather_tracks(self, exchange, ctx, local):
        allp, counts = self.ctx.allgather_tracks(local)
        return allp, [int(v) for v in counts]

    def close(self):
        pass  # the communicator goes with its context (ebo_destroy)


def bucket_rows(ev, cfg, row_b, row_e):
    """The events of grid rows [row_b, row_e) of one window, grouped by patch in grid order
    (time order kept inside a patch, as feature_detector.cpp:348-355 leaves them):
    -> (events, per-patch counts [rows * npx])."""
    iw, ih = cfg["image"]
    pw, ph = cfg["patch"]
    npx, npy = iw // pw, ih // ph
    gx = np.minimum(ev["x"] // pw, npx - 1)
    gy = np.minimum(ev["y"] // ph, npy - 1)
    sel = np.flatnonzero((gy >= row_b) & (gy < row_e))
    pid = (gy[sel] - row_b) * npx + gx[sel]
    order = np.argsort(pid, kind="stable")
    counts = np.bincount(pid, minlength=(row_e - row_b) * npx)
    return ev[sel[order]], counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("auto", "eval", "c4", "replicas"), default="auto")
    ap.add_argument("--config", type=int, default=3, help="eval / replicas: BASELINE config index (3 = configs[2])")
    ap.add_argument("--windows", type=int, default=None, help="eval / replicas: independent windows per GPU per step")
    ap.add_argument("--c4-windows", type=int, default=32, help="c4: windows in flight PER GPU (batch = N x this); "
                    "32 since round 4: one GPU's shard solves 8 / 16 / 32 windows in 1.73 / 1.55 / 1.45 ms per window "
                    "(8 192 units leave the chip's tail visible; 288 GB of HBM hold far more)")
    ap.add_argument("--strong", action="store_true", help="c4: keep the batch at --c4-windows windows in total")
    ap.add_argument("--no-c4-image", action="store_true", help="c4: stop the step at solve + all-gather (no final count image)")
    ap.add_argument("--replicas", action="store_true", help="same as --workload replicas (BASELINE configs[4])")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--comm", choices=("auto", "ebo", "torch"), default="torch",
                    help="N > 1 transport: torch = torch.distributed (backend nccl = RCCL; the default since round 5: the "
                         "library's grouped ncclSend / ncclRecv have never run between two ranks -- no box with two GPUs was "
                         "available -- and a deadlock inside the probe step is the one failure no fallback can rescue); "
                         "ebo = the library's own RCCL communicator (ebo_comm_init: ebo_allgather_device, "
                         "ebo_band_exchange_device, ebo_allgather_tracks); auto = ebo after one probed step and a vote, torch "
                         "when any rank says no (a rehearsal over gloo is always torch)")
    ap.add_argument("--c4-image", choices=("band", "dense"), default="band",
                    help="c4 final image: band = every rank keeps its own rows, only halo rows travel to the two neighbours "
                         "(SURVEY 8(e)); dense = a full image per window from every rank reduced onto rank 0 (round 3)")
    ap.add_argument("--c4-halo", type=int, default=32, help="c4 band image: halo rows above and below a rank's own rows")
    ap.add_argument("--preheat", type=int, default=None,
                    help="untimed steps BEFORE the W warm-up steps that bring the GPU's clocks up (the same count on every "
                         "rank: steps may hold collectives).  profiles/r04_step_loop.txt: the first ~10 ms of kernels after the "
                         "idle set-up phase run ~6 %% slower, whatever launches them")
    args = ap.parse_args()
    if args.replicas:
        args.workload = "replicas"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    workload = args.workload
    if workload == "auto":
        workload = "eval" if world == 1 else "c4"
    if args.steps is None:
        args.steps = {"eval": 200, "c4": 20, "replicas": 50}[workload]
    if args.warmup is None:
        args.warmup = {"eval": 10, "c4": 2, "replicas": 3}[workload]
    if args.windows is None:
        args.windows = 128 if args.config >= 3 else 256  # (C3: 64 / 128 / 256 windows per launch = 26.1 / 27.2 / 27.3 Gev/s)
    extras_on = rank == 0 and world == 1 and not args.no_extras and workload == "eval"

    # N > 1: the CPU baseline of rank 0 runs BEFORE anything touches the GPU or the process group (the
    # other ranks wait in the rendezvous), so that no rank sits in a collective while one computes
    base_early = None
    if world > 1 and rank == 0:
        try:
            synth_early = importlib.import_module("event-based-odomety_amd.synth")
            base_early = cpu_baseline(synth_early, 3 if workload == "c4" else args.config, args.cpu_seconds)
        except Exception as exc:
            base_early = {"error": repr(exc)}

    cpu_all = None
    if extras_on:
        try:
            cpu_all = cpu_baseline_all_cores(args.config, min(3.0, args.cpu_seconds))
        except Exception as exc:  # a reported extra, never a reason to lose the bench line
            cpu_all = {"error": repr(exc)}

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    rehearse = os.environ.get("EBO_BENCH_REHEARSE") == "1"
    if world > ndev and not rehearse:
        raise SystemExit("--gpus %d but only %d GPU(s) visible (EBO_BENCH_REHEARSE=1 shares them, for rehearsal only)"
                         % (world, ndev))
    dev = local % ndev if rehearse else local
    torch.cuda.set_device(dev)
    # EBO_BENCH_FORCE_DIST=1 takes the N > 1 code path (process group, barrier, reductions,
    # all-gathers) with a single rank: a rehearsal of that path on a one-GPU box
    force_dist = os.environ.get("EBO_BENCH_FORCE_DIST") == "1"
    ebo = importlib.import_module("event-based-odomety_amd")
    synth = importlib.import_module("event-based-odomety_amd.synth")
    exchange = importlib.import_module("event-based-odomety_amd.exchange")
    comm_notes = {}

    def make_comm(ctx_, probe=None):
        """The run's transport, chosen once the workload's context exists (the library's communicator lives on
        a context).  `auto`: the library's own communicator, checked with one barrier + one reduction and, when the
        workload gives one, `probe(candidate)` = one whole untimed step on it (every collective the timed steps will
        use, the grouped send / recv of the halo exchange among them); every rank votes (files, before any collective
        is relied on) and ALL ranks fall back to torch.distributed together when one of them could not do that."""
        want_ebo = (world > 1 or force_dist) and not rehearse and args.comm in ("auto", "ebo")
        if want_ebo:
            cand, ok, why = EboComm(torch, ebo, exchange, rank, world), True, None
            try:
                cand.attach(ctx_)
                cand.barrier()
                ok = cand.reduce(1.0, "SUM") == float(world)
                if ok and probe is not None:
                    probe(cand)
            except Exception as exc:
                ok, why = False, repr(exc)
            if args.comm == "ebo":
                if not ok:
                    raise SystemExit("--comm ebo: the library's communicator failed on rank %d: %s" % (rank, why))
                return cand
            try:
                agreed = exchange.agree(cand.prefix, rank, world, ok, timeout=COMM_TIMEOUT_S, not_before=T_START)
            except TimeoutError as exc:  # a rank that never voted is a `no` -- and every rank sees it as one
                agreed, why = False, "%s; %r" % (why, exc)
            if agreed:
                return cand
            comm_notes["fallback"] = "the library's communicator could not be set up and run one step on every rank (%s): torch.distributed" % (why,)
            try:
                ctx_.comm_destroy()
            except Exception:
                pass
        c_ = TorchComm(torch, rank, world, dev, rehearse, force_dist)
        c_.attach(ctx_)
        return c_

    comm = None
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        """average duration of fn's launches by HIP events on the stream they are launched on"""
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # ---------------------------------------------------------------------------------------
    # workloads: each returns (step, units-per-step fn, kernel timing fn, config dict, cleanup)
    # ---------------------------------------------------------------------------------------
    def setup_eval(cfg_idx, Wn, seq):
        """`Wn` windows of BASELINE config cfg_idx (sequence `seq`: distinct events per rank)."""
        cfg = synth.CONFIGS[cfg_idx]
        evs, gts = [], []
        for w in range(Wn):
            e, g = synth.make_window(cfg_idx, window=seq * Wn + w)
            evs.append(e)
            gts.append(g)
        offsets = np.zeros(Wn + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(e) for e in evs])
        ev = np.concatenate(evs)
        gt = np.stack(gts)
        ctx = ebo.Context(device=dev, image_w=cfg["image"][0], image_h=cfg["image"][1],
                          patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE,
                          grad=ebo.GRAD_JET, tv_weight=0.0, max_events=len(ev), max_windows=Wn)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_windows(ev, offsets)
        P = ctx.P
        active = sum(ctx.patch_info(p, w)[0] for w in range(Wn) for p in range(P) if ctx.patch_info(p, w)[1])
        d_flows = torch.from_numpy(gt * 0.5).to("cuda")  # mid-solve candidate flows
        d_out = torch.zeros((Wn * P, 3), dtype=torch.float64, device="cuda")
        return dict(cfg=cfg, ctx=ctx, ev=ev, offsets=offsets, gt=gt, P=P, active=active, n_events=len(ev),
                    d_flows=d_flows, d_out=d_out)

    def setup_c4(n_ranks, r, windows_total, distinct=4):
        """BASELINE configs[3] sharded by patch rows: rank r of n_ranks loads, for every window of the
        batch, the patches of ITS grid rows with the events inside them (ebo_set_patches)."""
        cfg = synth.CONFIGS[4]
        iw, ih = cfg["image"]
        pw, ph = cfg["patch"]
        npx, npy = iw // pw, ih // ph
        rows = exchange.shard_counts(npy, n_ranks)
        b, e = ebo.shard_range(npy, r, n_ranks)
        _, _, rects = synth.grid_rects(cfg["image"], cfg["patch"])
        my_rects = rects[b * npx:e * npx]
        # a few distinct windows, repeated to fill the batch (a window's events are generated by
        # numpy at ~2 s per 1 M events; each copy has its own buffers and is solved on its own)
        base, base_tref = [], []
        for w in range(min(distinct, windows_total)):
            ev_full = synth.make_window(4, window=w)[0]
            base.append(bucket_rows(ev_full, cfg, b, e))
            # the WINDOW's reference time (feature_detector.cpp:305-306): from its first / last event,
            # known to whoever cut the window; a shard cannot derive it from its own events
            base_tref.append(ebo.window_ref_time(ev_full["t_us"][0], ev_full["t_us"][-1]))
            del ev_full
        evs, cnts, t_ref = [], [], []
        for w in range(windows_total):
            ev_w, c_w = base[w % len(base)]
            evs.append(ev_w)
            cnts.append(c_w)
            t_ref.append(base_tref[w % len(base)])
        ev = np.concatenate(evs) if evs else np.zeros(0, dtype=ebo.EVENT_DTYPE)
        offs = np.zeros(windows_total * len(my_rects) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(np.concatenate(cnts))
        n_units = windows_total * len(my_rects)
        ctx = ebo.Context(device=dev, image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=ebo.LOSS_VARIANCE,
                          grad=ebo.GRAD_JET, tv_weight=0.0, max_events=max(len(ev), 1), max_windows=max(windows_total, 1))
        ctx.set_stream(stream.cuda_stream)
        ctx.set_patches(ev, offs, np.tile(my_rects, (windows_total, 1)))
        d_sol = torch.zeros((n_units, 2), dtype=torch.float64, device="cuda")
        d_stats = torch.zeros((n_units, 4), dtype=torch.int32, device="cuda")
        counts = [windows_total * q * npx for q in rows]
        d_all = torch.zeros((sum(counts), 2), dtype=torch.float64, device="cuda")
        n_ev_unit = np.diff(offs.astype(np.int64))
        # the final image: flows of ALL patches in patch order [window][P][2], partial image [window][H][W]
        d_grid = torch.zeros((windows_total, npx * npy, 2), dtype=torch.float64, device="cuda")
        # band-limited (default): the rank keeps its own image rows, halo rows go to the two neighbours.  Every rank
        # derives every rank's plan from the same row bounds, so "no plan" (halo larger than a neighbour) is the same
        # verdict everywhere and all ranks take the dense path together.
        bounds = [ebo.shard_range(npy, q, n_ranks)[0] * ph for q in range(n_ranks)] + [ih]
        band, B = None, None
        if not args.no_c4_image and args.c4_image == "band":
            try:
                band = ebo.band_plan(ih, bounds, r, args.c4_halo)
            except ebo.EboError:
                band = None
        if band is not None:
            mk = lambda rows_: torch.zeros((windows_total, rows_, iw), dtype=torch.int32, device="cuda") if rows_ else None
            B = dict(top=mk(band.top_rows), own=mk(band.own_rows), bottom=mk(band.bottom_rows),
                     from_above=mk(band.recv_above), from_below=mk(band.recv_below),
                     flag=torch.zeros(1, dtype=torch.int32, device="cuda"),
                     img_own=torch.zeros((windows_total, band.own_rows, iw), dtype=torch.float64, device="cuda"))
        dense = not args.no_c4_image and band is None
        d_img = torch.zeros((windows_total, ih, iw), dtype=torch.float64, device="cuda") if dense else None
        # the reduce of the partial images moves them as int32 (integer-valued counts): half the bytes, exact
        d_img32 = torch.zeros((windows_total, ih, iw), dtype=torch.int32, device="cuda") if (d_img is not None and n_ranks > 1) else None
        return dict(cfg=cfg, ctx=ctx, d_sol=d_sol, d_stats=d_stats, d_all=d_all, counts=counts, rows=rows,
                    t_ref=np.array(t_ref, dtype=np.int64), d_grid=d_grid, d_img=d_img, d_img32=d_img32, band=band, B=B,
                    bounds=bounds, iw=iw, ih=ih,
                    n_units=n_units, n_ev_unit=n_ev_unit, n_events=len(ev), npx=npx, npy=npy,
                    windows_total=windows_total, opts=ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT))

    def c4_event_evals(S):
        st = S["d_stats"].cpu().numpy()
        return float(((st[:, 1] + st[:, 2]).astype(np.int64) * S["n_ev_unit"]).sum())

    def c4_image(S_):
        """rank-major gathered blocks [rank][window][its rows][px] -> patch order [window][P][2], then the
        partial final image of this rank's events and ONE reduce onto rank 0"""
        at = 0
        col = 0
        for q, cnt_q in enumerate(S_["counts"]):
            nq = S_["rows"][q] * S_["npx"]
            if nq:
                S_["d_grid"][:, col:col + nq].copy_(S_["d_all"][at:at + cnt_q].view(S_["windows_total"], nq, 2))
            at += cnt_q
            col += nq
        if S_["band"] is not None:
            band, B = S_["band"], S_["B"]
            ptr = lambda t: t.data_ptr() if t is not None else 0
            S_["ctx"].count_image_band_device(S_["windows_total"], S_["t_ref"], S_["d_grid"].data_ptr(), band, ptr(B["top"]),
                                              ptr(B["own"]), ptr(B["bottom"]), B["flag"].data_ptr())
            comm.halo_exchange(exchange, S_["ctx"], S_["windows_total"], band, B)
            S_["ctx"].band_finish_device(S_["windows_total"], band, ptr(B["own"]), ptr(B["from_above"]), ptr(B["from_below"]),
                                         B["img_own"].data_ptr())
            return
        if S_["d_img"] is None:  # the band image escaped its halo: the dense image from here on
            S_["d_img"] = torch.zeros((S_["windows_total"], S_["ih"], S_["iw"]), dtype=torch.float64, device="cuda")
            S_["d_img32"] = torch.zeros((S_["windows_total"], S_["ih"], S_["iw"]), dtype=torch.int32, device="cuda") if world > 1 else None
        S_["ctx"].count_image_shard_device(S_["windows_total"], S_["t_ref"], S_["d_grid"].data_ptr(), S_["d_img"].data_ptr())
        comm.reduce_sum_to_rank0(S_["d_img"], S_.get("d_img32"))

    extras = {}
    roof_kernel = None
    if workload in ("eval", "replicas"):
        S = setup_eval(args.config, args.windows, rank)
        ctx, cfg = S["ctx"], S["cfg"]
        comm = make_comm(ctx)

        def eval_step():
            ctx.eval_device(S["d_flows"].data_ptr(), 1, S["d_out"].data_ptr())

        tracks_state = None
        if workload == "replicas":
            # the rank's tracked feature patches (the per-feature tracker objective, Optimizer::optimize):
            # a different number per sequence, so that the track lists are unequal
            rng = np.random.default_rng(7 + rank)
            H, Wd = cfg["image"][1], cfg["image"][0]
            ys, xs = np.mgrid[0:H, 0:Wd].astype(np.float64)
            img = sum(rng.uniform(-1, 1) * np.exp(-((xs - rng.uniform(0, Wd)) ** 2 + (ys - rng.uniform(0, H)) ** 2)
                                                  / (2 * rng.uniform(3, 9) ** 2)) for _ in range(40))
            gx, gy = np.zeros_like(img), np.zeros_like(img)
            gx[:, 1:-1] = 0.5 * (img[:, 2:] - img[:, :-2])
            gy[1:-1, :] = 0.5 * (img[2:, :] - img[:-2, :])
            co = ebo.Context(device=dev, image_w=Wd, image_h=H)
            co.optimizer_set_grad(gx, gy)
            n_tr = 100 - 3 * rank
            rects = np.stack([rng.uniform(5, Wd - 30, n_tr), rng.uniform(5, H - 30, n_tr),
                              np.full(n_tr, 25.0), np.full(n_tr, 25.0)], 1)
            nablas = [rng.integers(-3, 4, (25, 25)).astype(np.float64) for _ in range(n_tr)]
            tracks_state = dict(co=co, rects=rects, nablas=nablas, poses=np.tile([1.0, 0.0, 0.0, 0.0], (n_tr, 1)),
                                fds=rng.uniform(0, 6.28, n_tr), step=0, mine=np.zeros(0, dtype=ebo.TRACK_DTYPE),
                                gathered=0, counts=None)

        def step():
            eval_step()
            if tracks_state is not None:
                T = tracks_state
                poses, _, _ = T["co"].optimizer_solve(T["rects"], T["nablas"], T["poses"], T["fds"], normalize=True)
                pts = np.zeros(len(poses), dtype=ebo.TRACK_DTYPE)  # Patch::addTrajectoryPosition: new corner + time
                pts["id"] = np.arange(len(poses)) + 1000 * rank
                pts["t_us"] = 1_000_000 + 50_000 * T["step"]
                pts["x"] = T["rects"][:, 0] + 12.5 + poses[:, 2]
                pts["y"] = T["rects"][:, 1] + 12.5 + poses[:, 3]
                T["step"] += 1
                T["mine"] = pts  # the points this step added to the rank's tracks
                allp, counts = comm.allgather_tracks(exchange, ctx, pts)
                T["gathered"], T["counts"] = len(allp), counts

        units_per_step = float(S["active"])
        kernel_fn = eval_step
        roof_kernel = "k_eval3<true, false>"
        par = ("windows sharded over %d GPU(s), no data-path collective" % world) if workload == "eval" else (
            "%d independent sequences, one per GPU; per step one gather of the per-sequence tracks "
            "(counts + max-padded all-gather)" % world)
        config = {"workload": cfg["name"] + (" (BASELINE configs[4]: one sequence per GPU)" if workload == "replicas" else ""),
                  "windows_per_gpu_per_step": args.windows, "events_per_gpu_per_step": S["n_events"],
                  "scored_events_per_gpu_per_step": S["active"], "patches_per_window": S["P"],
                  "loss": "variance", "grad": "jet", "step": "one batched value+Jacobian evaluation"
                  + (" + tracker solve + track gather" if workload == "replicas" else ""), "parallelism": par,
                  "transport": comm.transport}
        config.update(comm_notes)
    else:
        wt = args.c4_windows if args.strong else args.c4_windows * world
        S = setup_c4(world, rank, wt)
        ctx, cfg = S["ctx"], S["cfg"]

        def solve_only():
            ctx.solve_device(S["opts"], S["d_sol"].data_ptr(), S["d_stats"].data_ptr())

        def step():
            solve_only()
            comm.allgather_rows(exchange, S["d_sol"], S["counts"], S["d_all"])
            if not args.no_c4_image:
                c4_image(S)

        def first_step_on(candidate):
            nonlocal comm
            comm = candidate
            step()
            torch.cuda.synchronize()

        comm = make_comm(ctx, probe=first_step_on)
        step()
        torch.cuda.synchronize()
        if S["band"] is not None and int(S["B"]["flag"].item()) != 0:
            # a solved flow could carry an event beyond the halo (the flag is the maximum over the ranks: the same
            # on every rank): this batch takes the dense image, as the product would
            comm_notes["band_escaped"] = "a unit's reach exceeded the halo of %d rows: dense image + reduce" % args.c4_halo
            S["band"] = None
            step()
            torch.cuda.synchronize()
        units_per_step = c4_event_evals(S)
        kernel_fn = solve_only
        roof_kernel = "k_solve_independent"
        config = {"workload": cfg["name"] + ", patch rows sharded over %d GPU(s)" % world,
                  "windows_in_batch": wt, "grid_rows_per_gpu": S["rows"], "patches_per_gpu_per_step": S["n_units"],
                  "events_per_gpu_per_step": S["n_events"], "event_evaluations_per_gpu_per_step": units_per_step,
                  "loss": "variance", "grad": "jet",
                  "event_evaluations": "events x the solver's evaluation requests (cost + Jacobian, from its statistics); a Jacobian "
                                       "request at the point whose cost was just evaluated runs the gather pass on the image that "
                                       "evaluation left in LDS (bit-identical result; the A/B build's EBO_SOLVE_NO_REUSE=1 rebuilds the image)",
                  "step": "device-resident per-patch solve of the shard (ebo_solve_device) + one all-gather of the solved flows"
                          + ("" if args.no_c4_image else (
                              " + band-limited final image: the shard's events counted into its own rows + a halo (ebo_count_image_band), "
                              "halo rows to the two neighbouring ranks (send / recv), own rows finished on the rank" if S["band"] is not None
                              else " + partial final count image of the shard's events (ebo_count_image_shard) + one reduce of the images onto rank 0")),
                  "transport": comm.transport,
                  "parallelism": "patch rows of every window sharded over %d GPU(s), one all-gather of flows (16 B/patch)%s per step"
                                 % (world, "" if args.no_c4_image else (
                                     " and one halo exchange with the neighbouring ranks" if S["band"] is not None else
                                     " and reduce of the %d x %d count images (moved as int32: exact, half the bytes of f64)" % (cfg["image"][0], cfg["image"][1])))}
        if not args.no_c4_image:
            dense_bytes = cfg["image"][0] * cfg["image"][1] * 8
            if S["band"] is not None:
                b_ = S["band"]
                sent = (b_.top_rows + b_.bottom_rows) * cfg["image"][0] * 4
                config["final_image"] = {"mode": "band", "halo_rows": args.c4_halo, "own_rows": b_.own_rows,
                                         "bytes_sent_per_rank_per_window": sent if world > 1 else 0,
                                         "dense_f64_image_bytes": dense_bytes, "fraction_of_dense": sent / dense_bytes,
                                         "note": "uint32 counts; interior ranks send both halos, the first and last rank one"}
            else:
                config["final_image"] = {"mode": "dense", "bytes_sent_per_rank_per_window": dense_bytes // 2 if world > 1 else 0,
                                         "dense_f64_image_bytes": dense_bytes}
        if world > 1:
            config["same_step_on_one_gpu"] = ("the N = 1 line's `value` is another workload (one batched evaluation of configs[2]); THIS step "
                                              "on one GPU is that line's extras.c4_sharded_solve_step_1gpu.mevents_per_s (or `--gpus 1 --workload c4`), the "
                                              "reference a scaling efficiency of this line has to be taken against")
        config.update(comm_notes)

    # ---- the timed region: W warm-up steps, barrier, EXACTLY K steps, barrier -----------------
    # Before it, untimed: the GPU idled through the set-up (event generation, uploads) and its first ~10 ms of
    # kernels run at lower clocks.  With the driver's `--steps 20 --warmup 5` that ramp was 6 % of round 3's
    # headline (ms_per_step 0.568 against a kernel of 0.533 ms by HIP events in the same run).
    if args.preheat is None:
        args.preheat = {"eval": 200, "c4": 6, "replicas": 20}[workload]
    for _ in range(args.preheat):
        step()
    for _ in range(args.warmup):
        step()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    comm.barrier()
    dt = time.perf_counter() - t0
    dt = comm.reduce(dt, "MAX")
    HEADLINE_DT = dt  # the line at the end uses THIS name: nothing the extras compute may shadow the timed region
    total_units = comm.reduce(units_per_step, "SUM")
    comm_total_events = comm.reduce(float(S["n_events"]), "SUM") if workload == "c4" else None
    band_counted, band_full = None, None
    if workload == "c4" and S.get("band") is not None:
        # untimed checks of the band image: every rank's own rows sum to the events counted; on the library's
        # communicator the whole image is also assembled once on rank 0 (ebo_band_gather_device: owned rows only)
        band_counted = comm.reduce(float(S["B"]["img_own"].sum().item()), "SUM")
        band_escaped = comm.reduce(float(S["B"]["flag"].item()), "MAX")
        if isinstance(comm, EboComm) or world == 1:
            d_full = torch.zeros((S["windows_total"], S["ih"], S["iw"]), dtype=torch.float64, device="cuda") if rank == 0 else None
            ctx.band_gather_device(S["windows_total"], S["bounds"], S["B"]["img_own"].data_ptr(), 0,
                                   d_full.data_ptr() if d_full is not None else 0)
            torch.cuda.synchronize()
            if rank == 0:
                full = d_full.cpu().numpy()
                band_full = {"assembled_on_rank0": True, "integer_valued": bool(np.array_equal(full, np.round(full))),
                             "sum_equals_own_rows_sum": bool(full.sum() == band_counted)}

    # ---- dominant kernel: average launch duration by HIP events on ITS stream -----------------
    kern_ms = timed(kernel_fn, max(3, args.steps))
    lds_rates = None
    if rank == 0 and workload != "c4":
        try:
            lds_rates = ctx.lds_rates()  # the LDS rates the kernel is priced against, measured in this run
        except Exception:
            lds_rates = None
    achieved = BYTES_PER_EVENT_EVAL * units_per_step / (kern_ms * 1e-3) / 1e9

    if workload == "replicas" and rank == 0:
        T = tracks_state
        extras["track_gather"] = {"records_per_step_all_ranks": T["gathered"], "per_rank": T["counts"]}
    if workload == "c4" and rank == 0:
        # the gathered buffer holds every window's flows: rank q's block = its rows of every window
        got = S["d_all"].cpu().numpy()
        mine = S["d_sol"].cpu().numpy()
        lo = sum(S["counts"][:rank])
        extras["allgather_check"] = bool(np.array_equal(got[lo:lo + len(mine)], mine)) and bool(np.isfinite(got).all())
        st = S["d_stats"].cpu().numpy()
        extras["mean_evals_per_patch"] = float((st[:, 1] + st[:, 2])[st[:, 2] > 0].mean())
        if S.get("band") is not None:
            extras["final_image"] = {"mode": "band", "windows": int(S["windows_total"]), "events_counted": band_counted,
                                     "events_in_batch": float(comm_total_events) if comm_total_events is not None else None,
                                     "escaped_flag_max_over_ranks": band_escaped}
            if band_full:
                extras["final_image"].update(band_full)
        elif S["d_img"] is not None:
            img = S["d_img"].cpu().numpy()
            # every event of every window of the batch lands at most once: integer-valued, sum <= events
            extras["final_image"] = {"windows": int(img.shape[0]), "integer_valued": bool(np.array_equal(img, np.round(img))),
                                     "events_counted": float(img.sum()),
                                     "events_in_batch": float(comm_total_events) if comm_total_events is not None else None}

    if cpu_all is not None:
        extras["cpu_baseline_all_cores"] = cpu_all
    if extras_on:  # single-GPU diagnostics; ranks of an N > 1 run stay in step
        Wn, P, n_events, offsets, ev = args.windows, S["P"], S["n_events"], S["offsets"], S["ev"]
        d_flows, d_out = S["d_flows"], S["d_out"]

        def rate(n, ms):
            return n / (ms * 1e-3) / 1e6

        # value-only evaluation (the cost-only evaluations of the LM loop)
        ms = timed(lambda: ctx.eval_device(d_flows.data_ptr(), 0, d_out.data_ptr()), args.steps)
        extras["value_only_mevents_per_s"] = rate(S["active"], ms)
        # device-resident independent solve of every window (one launch)
        d_sol = torch.zeros((Wn * P, 2), dtype=torch.float64, device="cuda")
        d_stats = torch.zeros((Wn * P, 4), dtype=torch.int32, device="cuda")
        opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT)
        ctx.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr())
        solve_ms = timed(lambda: ctx.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr()), 2)
        st = d_stats.cpu().numpy().reshape(Wn, P, 4)
        ne = np.array([[ctx.patch_info(p, w)[0] for p in range(P)] for w in range(Wn)])
        evals = (st[:, :, 1] + st[:, :, 2]) * ne
        extras["solve_independent"] = {
            "ms": solve_ms, "windows": Wn, "mevents_per_s": rate(float(evals.sum()), solve_ms),
            "mean_evals_per_patch": float((st[:, :, 1] + st[:, :, 2])[st[:, :, 2] > 0].mean()),
            "mean_jacobian_evals_per_patch": float(st[:, :, 2][st[:, :, 2] > 0].mean()),
            "note": "evaluations = the solver's requests; a Jacobian request at the point whose cost was just evaluated "
                    "reuses the image in LDS (same bits)"}
        # single-window latency (one window, one launch)
        c1 = ebo.Context(device=dev, image_w=cfg["image"][0], image_h=cfg["image"][1],
                         patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE,
                         tv_weight=0.0, max_events=int(offsets[1]), max_windows=1)
        c1.set_stream(stream.cuda_stream)
        c1.set_window(ev[: int(offsets[1])])
        f1 = d_flows[:1].contiguous()
        o1 = torch.zeros((P, 3), dtype=torch.float64, device="cuda")
        for _ in range(5):
            c1.eval_device(f1.data_ptr(), 1, o1.data_ptr())
        extras["single_window_eval_us"] = timed(lambda: c1.eval_device(f1.data_ptr(), 1, o1.data_ptr()), 200) * 1e3
        c1.close()
        # window set-up: host counting sort + 8 B/event upload  vs  24 B/event upload + device
        # bucketing  vs  device bucketing of events already resident (ebo_set_windows_device)
        # (the host-bucketing and the every-evaluation-rebuilds-its-image legs of rounds 1-3 were A/B of superseded
        # paths: they live in tools/time_ingest.py / tools/ab_solve.py on the -DEBO_AB build now)

        def best_of(fn, reps=3):
            """steady-state time of one call: a warm-up, then the best of `reps` (a single shot of a
            2 ms call read 2-4x slow every few runs: first use of the copy stream, host scheduling)"""
            fn()
            best = None
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                dt_ = time.perf_counter() - t0
                best = dt_ if best is None else min(best, dt_)
            return best

        t_dev = best_of(lambda: ctx.set_windows(ev, offsets))
        d_raw = torch.from_numpy(ev.view(np.uint8).reshape(-1, 24)).to("cuda")
        torch.cuda.synchronize()
        t_res = best_of(lambda: ctx.set_windows_device(d_raw.data_ptr(), offsets))
        # compact 8-byte records (ebo_set_windows8) from page-locked host memory: a third of the bytes
        # over PCIe, upload pipelined with the bucketing in groups of windows
        t_base = np.array([int(ev["t_us"][int(offsets[w])]) for w in range(Wn)], dtype=np.int64)
        pin8 = torch.empty((n_events, 8), dtype=torch.uint8).pin_memory()
        v8 = pin8.numpy().view(ebo.EVENT8_DTYPE).reshape(-1)
        for w in range(Wn):
            a, b = int(offsets[w]), int(offsets[w + 1])
            ebo.pack_events8(ev[a:b], t_base[w], out=v8[a:b])
        t_c8 = best_of(lambda: ctx.set_windows8(pin8.data_ptr(), t_base, offsets))
        pin24 = torch.from_numpy(ev.view(np.uint8).reshape(-1, 24)).pin_memory()
        ev24p = pin24.numpy().view(ebo.EVENT_DTYPE).reshape(-1)
        t_p24 = best_of(lambda: ctx.set_windows(ev24p, offsets))
        extras["window_setup_mevents_per_s"] = {
            "raw_upload_plus_device_bucketing": n_events / t_dev / 1e6,
            "raw_upload_pinned_plus_device_bucketing": n_events / t_p24 / 1e6,
            "compact8_upload_pinned_plus_device_bucketing": n_events / t_c8 / 1e6,
            "device_bucketing_resident_events": n_events / t_res / 1e6}
        del d_raw, pin8, pin24

        # text ingest: a DAVIS events.txt of the window's events, parsed by the library on the host's
        # threads (csrc/txt_events.h; the reference reads with hardware_concurrency() threads as well) -- lines per
        # second with one thread and with the default thread count, and the thread count that parsed
        try:
            import tempfile
            n_lines = 2_000_000
            rng_t = np.random.default_rng(11)
            tcol = 1468941032.0 + np.cumsum(rng_t.integers(0, 50, n_lines)) * 1e-6
            xs, ys, ss = rng_t.integers(0, 346, n_lines), rng_t.integers(0, 260, n_lines), rng_t.integers(0, 2, n_lines)
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "events.txt")
                with open(path, "w") as fh:
                    for lo in range(0, n_lines, 200_000):
                        hi = lo + 200_000
                        fh.write("".join("%.6f %d %d %d\n" % q for q in zip(tcol[lo:hi], xs[lo:hi], ys[lo:hi], ss[lo:hi])))
                buf = np.zeros(n_lines, dtype=ebo.EVENT_DTYPE)
                res = {}
                for label_t, thr in (("one_thread", 1), ("default_threads", 0)):
                    best, used = None, 1
                    for _ in range(3):
                        t_in = time.perf_counter()
                        got, _, used = ebo.read_events_txt_threads(path, n_lines, thr, out=buf)
                        t_in = time.perf_counter() - t_in  # (not `dt`: that is the timed region's, used by the line below)
                        best = t_in if best is None else min(best, t_in)
                    assert len(got) == n_lines
                    res[label_t] = {"mlines_per_s": n_lines / best / 1e6, "threads": used}
                res["file_mb"] = os.path.getsize(path) / 1e6
                res["lines"] = n_lines
                res["host_cpus"] = len(os.sched_getaffinity(0))
            extras["text_ingest_mlines_per_s"] = res
        except Exception as exc:  # a reported extra
            extras["text_ingest_mlines_per_s"] = {"error": repr(exc)}

        def eval_extra(ci, wn, loss, label, reps=10, cfgd=None):
            """value+Jacobian evaluation of `wn` windows of config ci with `loss`"""
            c = cfgd or synth.CONFIGS[ci]
            evx, offx, gtx = synth.make_stream(ci if cfgd is None else cfgd, wn)
            cx = ebo.Context(device=dev, image_w=c["image"][0], image_h=c["image"][1], patch_w=c["patch"][0],
                             patch_h=c["patch"][1], loss=loss, tv_weight=0.0, max_events=len(evx), max_windows=wn)
            cx.set_stream(stream.cuda_stream)
            cx.set_windows(evx, offx)
            fx = torch.from_numpy(gtx * 0.5).to("cuda")
            ox = torch.zeros((wn * cx.P, 3), dtype=torch.float64, device="cuda")
            # steady state: a new context's set-up leaves the GPU idle for milliseconds and the clock takes a few
            # hundred milliseconds of work to come back (the first launches after it read ~5 % slow): warm up for
            # ~100 ms, then the best of three batches
            for _ in range(50):
                cx.eval_device(fx.data_ptr(), 1, ox.data_ptr())
            msx = min(timed(lambda: cx.eval_device(fx.data_ptr(), 1, ox.data_ptr()), reps) for _ in range(3))
            extras[label] = {"ms": msx, "windows": wn, "patches_per_window": cx.P, "mevents_per_s": rate(len(evx), msx)}
            if loss == ebo.LOSS_EDGE:
                # the roofline of the reference's ACTIVE loss: f64 vector arithmetic (not HBM, not MFMA).  The
                # work is counted by the kernel itself in one extra evaluation (ebo_edge_work_stats)
                msv = min(timed(lambda: cx.eval_device(fx.data_ptr(), 0, ox.data_ptr()), reps) for _ in range(3))
                st = cx.edge_work_stats(fx.data_ptr(), True)
                fj, fv = edge_flops(st, True), edge_flops(st, False)
                extras[label]["value_only_ms"] = msv
                extras[label]["edge_roofline"] = {
                    "bound": "f64 vector ALU", "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "work": st,
                    "flops_per_item": EDGE_FLOPS, "useful_flops_with_jacobian": fj, "useful_flops_value_only": fv,
                    "achieved_with_jacobian": fj / (msx * 1e-3) / 1e12, "achieved_value_only": fv / (msv * 1e-3) / 1e12,
                    "frac_with_jacobian": fj / (msx * 1e-3) / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                    "frac_value_only": fv / (msv * 1e-3) / 1e12 / F64_VECTOR_PEAK_TFLOPS}
            cx.close()

        # the other BASELINE sizes with the same kernel, and the reference's own default objective
        # (edge / structure-tensor loss) on C2 and on the reference-default grid
        eval_extra(2, 256, ebo.LOSS_VARIANCE, "c2_240x180_value_jacobian")
        eval_extra(4, 8, ebo.LOSS_VARIANCE, "c4_1280x720_value_jacobian")
        eval_extra(2, 64, ebo.LOSS_EDGE, "edge_loss_value_jacobian")
        eval_extra(3, 16, ebo.LOSS_EDGE, "edge_loss_value_jacobian_c3")
        rcfg = dict(name="reference default", image=(240, 180), patch=(20, 20), events=15000, index=0)
        eval_extra(0, 256, ebo.LOSS_EDGE, "edge_loss_value_jacobian_reference_default", cfgd=rcfg)

        # the per-patch (TV-free) solve with the reference's own loss: device-resident (k_solve_edge, the
        # default) against host LMs in lock step over batched evaluations, 256 reference-default windows
        evs_, offs_, _ = synth.make_stream(rcfg, 256)
        ce = ebo.Context(device=dev, image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE, tv_weight=0.0,
                         max_events=len(evs_), max_windows=256)
        ce.set_windows(evs_, offs_)
        es = {}
        ce.solve(ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT))
        t0 = time.perf_counter()
        _, ss_ = ce.solve(ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT))
        es["device_ms"] = (time.perf_counter() - t0) * 1e3
        es["windows"] = 256
        es["evaluations_window0"] = int(ss_[0].num_evals_cost + ss_[0].num_evals_jac)
        extras["edge_solve_independent_reference_default"] = es
        ce.close()
        del evs_, offs_

        # integer count images (the HBM-bound kernels) on working sets >= 1 GiB: a few distinct
        # windows repeated (every copy has its own events and its own image in HBM)
        def count_extra(ci, distinct, copies, label):
            c = synth.CONFIGS[ci]
            evx, offx, gtx = synth.make_stream(ci, distinct)
            wn = distinct * copies
            ev_all = np.tile(evx, copies)
            off_all = np.zeros(wn + 1, dtype=np.uint64)
            per = np.diff(offx.astype(np.int64))
            off_all[1:] = np.cumsum(np.tile(per, copies))
            cx = ebo.Context(device=dev, image_w=c["image"][0], image_h=c["image"][1], patch_w=c["patch"][0],
                             patch_h=c["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev_all),
                             max_windows=wn)
            cx.set_stream(stream.cuda_stream)
            cx.set_windows(ev_all, off_all)
            n_ev = len(ev_all)
            del ev_all
            fl = torch.from_numpy(np.tile(gtx, (copies, 1, 1))).to("cuda")  # ground-truth flows = a solved window's
            d_img = torch.zeros((wn, c["image"][1], c["image"][0]), dtype=torch.float64, device="cuda")
            nbytes = 8 * n_ev + d_img.numel() * 8
            res = {"windows": wn, "working_set_bytes": nbytes}
            for mode, name, aux in ((ebo.COUNT_WARPED, "warped", fl.data_ptr()), (ebo.COUNT_INTEGRATED, "integrated", 0)):
                cx.count_image_device(mode, aux, d_img.data_ptr())
                msx = min(timed(lambda: cx.count_image_device(mode, aux, d_img.data_ptr()), 10) for _ in range(3))
                res[name] = {"ms": msx, "mevents_per_s": rate(n_ev, msx), "gbs_algorithmic": nbytes / (msx * 1e-3) / 1e9,
                             "hbm_frac": nbytes / (msx * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # same-run yardstick: the same bytes (events read once + image written once) with no work
            moved = cx.stream_yardstick_device(d_img.data_ptr())
            msy = min(timed(lambda: cx.stream_yardstick_device(d_img.data_ptr()), 10) for _ in range(3))
            res["plain_stream_same_bytes"] = {"ms": msy, "gbs": moved / (msy * 1e-3) / 1e9,
                                              "hbm_frac": moved / (msy * 1e-3) / 1e9 / HBM_PEAK_GBS}
            for name in ("warped", "integrated"):
                res[name]["frac_of_plain_stream"] = msy / res[name]["ms"]
            # The warped kernel reads a unit within reach of two tiles twice: its REAL traffic (PMC counters of the same
            # command, profiles/traffic.json) over the algorithmic bytes, and the rate it moves that traffic at against the
            # plain stream's -- the review's alternative bar for this kernel (>= 0.95: what is left is the re-read).
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as fp:
                    for k_ in json.load(fp)["kernels"]:
                        wl = k_.get("workload", "")
                        if k_.get("kernel") == "k_count_tiles" and wl.startswith("C%d " % ci) and ("%d windows" % wn) in wl:
                            ratio = k_["hbm_bytes_per_launch"] / k_["algorithmic_bytes_per_launch"]
                            res["warped"]["traffic_over_algorithmic"] = ratio
                            res["warped"]["real_traffic_rate_over_plain_stream"] = res["warped"]["frac_of_plain_stream"] * ratio
                            res["warped"]["traffic_source"] = k_.get("source")
            except (OSError, KeyError, ValueError):
                pass
            extras[label] = res
            cx.close()
            del d_img, fl

        count_extra(2, 128, 12, "count_image_c2")   # 1536 windows: 1.15 GB
        count_extra(3, 32, 16, "count_image_c3")    # 512 windows: 1.19 GB
        count_extra(4, 4, 18, "count_image_c4")     # 72 windows: 1.11 GB

        # the N > 1 default workload on this one GPU (the same step: whole grid = one shard, a
        # self-copy in place of the all-gather): the per-GPU reference the scaling runs compare with
        S4 = setup_c4(1, 0, args.c4_windows)

        def c4_step(image=True):
            S4["ctx"].solve_device(S4["opts"], S4["d_sol"].data_ptr(), S4["d_stats"].data_ptr())
            S4["d_all"].copy_(S4["d_sol"])
            if image and not args.no_c4_image:
                c4_image(S4)

        c4_step()
        ms4 = timed(c4_step, 5)
        ms4_solve = timed(lambda: c4_step(False), 5)
        ee = c4_event_evals(S4)
        extras["c4_sharded_solve_step_1gpu"] = {"ms": ms4, "ms_without_final_image": ms4_solve, "windows": args.c4_windows,
                                                "event_evaluations": ee, "mevents_per_s": rate(ee, ms4),
                                                "final_image": "band" if S4["band"] is not None else "dense"}
        S4["ctx"].close()
        del S4

        # the reference's own call (240x180, 20x20 patches, 15 k events, edge loss, TV-coupled
        # global LM: FeatureDetector::compensateEventsContrast as shipped), 64 windows in lock step
        rev, roff, _ = synth.make_stream(rcfg, 64)
        cr = ebo.Context(device=dev, image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE,
                         max_events=len(rev), max_windows=64)
        cr.set_windows(rev, roff)
        cr.solve(ebo.default_solver())
        t0 = time.perf_counter()
        _, rs = cr.solve(ebo.default_solver())
        t_ref = time.perf_counter() - t0
        cr.close()
        # ... and one window alone (what tools::Evaluator's per-window call sees)
        cr1 = ebo.Context(device=dev, image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE,
                          max_events=int(roff[1]), max_windows=1)
        cr1.set_windows(rev[:int(roff[1])], roff[:2])
        cr1.solve(ebo.default_solver())
        t_one = None
        for _ in range(3):
            t0 = time.perf_counter()
            cr1.solve(ebo.default_solver())
            t1_ = time.perf_counter() - t0
            t_one = t1_ if t_one is None else min(t_one, t1_)
        cr1.close()
        if "edge_loss_value_jacobian_reference_default" in extras:
            extras["edge_roofline"] = extras["edge_loss_value_jacobian_reference_default"].get("edge_roofline")
        extras["reference_default_call"] = {"windows": 64, "ms_per_window": t_ref * 1e3 / 64,
                                            "iterations": rs[0].iterations, "ms_single_window": t_one * 1e3}
        # per-feature tracker objective (Optimizer::optimize's solve), 100 tracked 25x25 patches
        rng = np.random.default_rng(7)
        ys, xs = np.mgrid[0:180, 0:240].astype(np.float64)
        img = sum(rng.uniform(-1, 1) * np.exp(-((xs - rng.uniform(0, 240)) ** 2 + (ys - rng.uniform(0, 180)) ** 2)
                                              / (2 * rng.uniform(3, 9) ** 2)) for _ in range(40))
        gx, gy = np.zeros_like(img), np.zeros_like(img)
        gx[:, 1:-1] = 0.5 * (img[:, 2:] - img[:, :-2])
        gy[1:-1, :] = 0.5 * (img[2:, :] - img[:-2, :])
        co = ebo.Context(device=dev, image_w=240, image_h=180)
        co.optimizer_set_grad(gx, gy)
        rects = np.stack([rng.uniform(5, 210, 100), rng.uniform(5, 150, 100), np.full(100, 25.0), np.full(100, 25.0)], 1)
        nablas = [rng.integers(-3, 4, (25, 25)).astype(np.float64) for _ in range(100)]
        poses = np.tile([1.0, 0.0, 0.0, 0.0], (100, 1))
        fds = rng.uniform(0, 6.28, 100)
        co.optimizer_solve(rects, nablas, poses, fds, normalize=True)
        t0 = time.perf_counter()
        co.optimizer_solve(rects, nablas, poses, fds, normalize=True)
        extras["tracker_optimizer_solve"] = {"patches": 100, "ms": (time.perf_counter() - t0) * 1e3}
        co.close()

    if rank == 0:
        base = cpu_baseline(synth, args.config if workload != "c4" else 3, args.cpu_seconds) if world == 1 else base_early
        # HBM traffic of the dominant kernel per launch, from the committed PMC profile of
        # this same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            for t in json.load(open(tpath)).get("kernels", []):
                same = (t.get("c4_windows_per_gpu") == args.c4_windows) if workload == "c4" else (
                    t.get("windows") == config.get("windows_per_gpu_per_step"))
                if t.get("kernel") == roof_kernel and t.get("workload") == cfg["name"] and same:
                    traffic, traffic_src = t.get("hbm_bytes_per_launch"), t.get("source")
        value = total_units * args.steps / HEADLINE_DT / 1e6
        line = {
            "metric": "Mevents/s warped+scored (value+Jacobian of the variance-contrast objective)"
                      if workload != "c4" else "Mevents/s warped+scored (event-evaluations of the per-patch solves)",
            "value": value, "unit": "Mevents/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "preheat_steps_untimed": args.preheat, "ms_per_step": HEADLINE_DT / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (workload == "c4" and args.strong) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": config,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": roof_kernel, "kernel_ms": kern_ms,
                         "note": "algorithmic 8 B/event-evaluation; the kernel is bound by LDS "
                                 "atomics + f64 VALU, not HBM (DESIGN.md section 4)",
                         # what does bind it: 49 64-bit LDS atomics + 49 64-bit LDS reads per event-evaluation
                         # against the chip-wide rates of tools/microbench/lds_atomics.hip on this GPU
                         # (ds_add_u64 and ds_read_b64 at random addresses)
                         "lds": lds_roofline(units_per_step, kern_ms, lds_rates) if workload != "c4" else None},
            "cpu_baseline": base,
            "extras": extras,
        }
        if rehearse or (comm.active and world == 1):
            line["rehearsal"] = "N > 1 code path on shared GPU(s) over %s: not a scaling measurement" % (comm.backend,)
        print(json.dumps(line), flush=True)
    ctx.close()
    comm.close()


if __name__ == "__main__":
    main()
