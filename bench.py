#!/usr/bin/env python3
"""bench.py — Mevents/s warped+scored on MI355X (BASELINE.json's metric).

A step = ONE batched objective evaluation (value + Jacobian of the variance-contrast
objective, EBO_GRAD_JET) of every patch of `--windows` independent windows of the
BASELINE config (default configs[1]: 240x180, 64 patches, 50k events/window), i.e.
every event is warped by its patch's candidate flow, splatted and scored once per
step.  Events, patch table and flows are resident in HBM before the timed region.
Windows are independent problems (the reference re-initialises the flow to 0 for each
window, feature_detector.cpp:318-326), so a stream is evaluated many windows at a time.

N > 1 (torchrun, one rank per GPU): the windows (units) are sharded over ranks and each
rank evaluates its own shard.  Whole windows are independent problems, so the data path
has NO collective; RCCL carries only the barrier and the max-over-ranks of the timing.
(`--exchange` adds the one exchange the path has when a SINGLE window is sharded over
GPUs — C4's 128 patches per GPU: an all-gather of the (r, J0, J1) triples per step for the
replicated host solver)  Per-GPU work is fixed as N grows => weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_EVENT_EVAL = 8  # packed {x:15, pol:1, y:15, dt:32}; SURVEY §8(d)


def cpu_baseline(synth, cfg_idx, seconds_budget):
    """The oracle (CPU port of the reference path, single thread like the reference)
    timed on this box's host cores on a bounded sample of the same workload."""
    import orc
    cfg = synth.CONFIGS[cfg_idx]
    ev, gt = synth.make_window(cfg_idx)
    prm = orc.default_params(image_w=cfg["image"][0], image_h=cfg["image"][1],
                             patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=1, tv_weight=0.0)
    flows = gt * 0.5
    sec, n = orc.window_eval_timed(ev, prm, flows, True, 1)  # warm-up + calibration
    reps = max(1, min(2000, int(seconds_budget / max(sec, 1e-6))))
    sec, n = orc.window_eval_timed(ev, prm, flows, True, reps)
    return {
        "value": n / sec / 1e6, "unit": "Mevents/s", "cores": 1, "kind": "port",
        "sample": "%d value+Jacobian evaluations of 1 window of %s (%d event-evaluations, %.1f s) "
                  "by oracle/liboracle.so, single thread like the reference" % (reps, cfg["name"], n, sec),
    }


def _cpu_worker(args):
    """One host core: value+Jacobian evaluations of its own synthetic window by the oracle."""
    cfg_idx, window, seconds = args
    import importlib
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, help="BASELINE config index (2 = configs[1])")
    ap.add_argument("--windows", type=int, default=256, help="independent windows per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--exchange", action="store_true",
                    help="N > 1: all-gather (r, J0, J1) after every step (a window sharded over GPUs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_all = None
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            cpu_all = cpu_baseline_all_cores(args.config, min(3.0, args.cpu_seconds))
        except Exception as exc:  # a reported extra, never a reason to lose the bench line
            cpu_all = {"error": repr(exc)}

    import torch
    import torch.distributed as dist

    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    # EBO_BENCH_FORCE_DIST=1 takes the N > 1 code path (process group, barrier, reductions) with a
    # single rank: a rehearsal of that path on a one-GPU box
    use_dist = world > 1 or os.environ.get("EBO_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))

    ebo = importlib.import_module("event-based-odomety_amd")
    synth = importlib.import_module("event-based-odomety_amd.synth")
    cfg = synth.CONFIGS[args.config]

    # ---- synthetic stream: `windows` windows per rank, distinct per rank ---------
    Wn = args.windows
    evs, gts = [], []
    for w in range(Wn):
        e, g = synth.make_window(args.config, window=rank * Wn + w)
        evs.append(e)
        gts.append(g)
    offsets = np.zeros(Wn + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    ev = np.concatenate(evs)
    n_events = int(len(ev))
    gt = np.stack(gts)
    del evs

    ctx = ebo.Context(device=local, image_w=cfg["image"][0], image_h=cfg["image"][1],
                      patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE,
                      grad=ebo.GRAD_JET, tv_weight=0.0, max_events=n_events, max_windows=Wn)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_windows(ev, offsets)
    P = ctx.P
    active_events = sum(ctx.patch_info(p, w)[0] for w in range(Wn) for p in range(P) if ctx.patch_info(p, w)[1])

    d_flows = torch.from_numpy(gt * 0.5).to("cuda")  # mid-solve candidate flows
    d_out = torch.zeros((Wn * P, 3), dtype=torch.float64, device="cuda")
    exchange = use_dist and args.exchange
    d_all = torch.zeros((world * Wn * P, 3), dtype=torch.float64, device="cuda") if exchange else None

    def step():
        ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())
        if exchange:
            dist.all_gather_into_tensor(d_all, d_out)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([active_events], dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_events = float(cnt.item())
    else:
        total_events = float(active_events)

    # ---- dominant kernel: average launch duration by HIP events on ITS stream -----
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(args.steps):
        ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / args.steps
    achieved = BYTES_PER_EVENT_EVAL * active_events / (kern_ms * 1e-3) / 1e9

    extras = {}
    if cpu_all is not None:
        extras["cpu_baseline_all_cores"] = cpu_all
    if rank == 0 and world == 1 and not args.no_extras:  # single-GPU diagnostics; ranks of an N > 1 run stay in step
        # value-only evaluation (the cost-only evaluations of the LM loop)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(args.steps):
            ctx.eval_device(d_flows.data_ptr(), 0, d_out.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        extras["value_only_mevents_per_s"] = active_events / (e0.elapsed_time(e1) / args.steps * 1e-3) / 1e6
        # device-resident independent solve of every window (one launch)
        d_sol = torch.zeros((Wn * P, 2), dtype=torch.float64, device="cuda")
        d_stats = torch.zeros((Wn * P, 4), dtype=torch.int32, device="cuda")
        opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT)
        ctx.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        ctx.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        st = d_stats.cpu().numpy().reshape(Wn, P, 4)
        ne = np.array([[ctx.patch_info(p, w)[0] for p in range(P)] for w in range(Wn)])
        evals = (st[:, :, 1] + st[:, :, 2]) * ne
        solve_ms = e0.elapsed_time(e1)
        extras["solve_independent"] = {
            "ms": solve_ms, "windows": Wn, "mevents_per_s": float(evals.sum()) / (solve_ms * 1e-3) / 1e6,
            "mean_evals_per_patch": float((st[:, :, 1] + st[:, :, 2])[st[:, :, 2] > 0].mean()),
        }
        # integer count image (HBM-bound kernel): warped by the solved flows
        d_img = torch.zeros((Wn, cfg["image"][1], cfg["image"][0]), dtype=torch.float64, device="cuda")
        ctx.count_image_device(ebo.COUNT_WARPED, d_sol.data_ptr(), d_img.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(10):
            ctx.count_image_device(ebo.COUNT_WARPED, d_sol.data_ptr(), d_img.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        cms = e0.elapsed_time(e1) / 10
        img_bytes = d_img.numel() * 8
        extras["count_image_warped"] = {
            "ms": cms, "mevents_per_s": n_events / (cms * 1e-3) / 1e6,
            "gbs_algorithmic": (8 * n_events + img_bytes) / (cms * 1e-3) / 1e9,
            "hbm_frac": (8 * n_events + img_bytes) / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        # un-warped count image (integrateEvents, feature_detector.cpp:466-482)
        ctx.count_image_device(ebo.COUNT_INTEGRATED, 0, d_img.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(10):
            ctx.count_image_device(ebo.COUNT_INTEGRATED, 0, d_img.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        cms = e0.elapsed_time(e1) / 10
        extras["count_image_integrated"] = {
            "ms": cms, "mevents_per_s": n_events / (cms * 1e-3) / 1e6,
            "gbs_algorithmic": (8 * n_events + img_bytes) / (cms * 1e-3) / 1e9,
            "hbm_frac": (8 * n_events + img_bytes) / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        # single-window latency (one 50k-event window, one launch)
        c1 = ebo.Context(device=local, image_w=cfg["image"][0], image_h=cfg["image"][1],
                         patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE,
                         tv_weight=0.0, max_events=int(offsets[1]), max_windows=1)
        c1.set_stream(stream.cuda_stream)
        c1.set_window(ev[: int(offsets[1])])
        f1 = d_flows[:1].contiguous()
        o1 = torch.zeros((P, 3), dtype=torch.float64, device="cuda")
        for _ in range(5):
            c1.eval_device(f1.data_ptr(), 1, o1.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(200):
            c1.eval_device(f1.data_ptr(), 1, o1.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        extras["single_window_eval_us"] = e0.elapsed_time(e1) / 200 * 1e3
        c1.close()
        # window set-up: host counting sort + 8 B/event upload  vs  24 B/event upload + device
        # bucketing  vs  device bucketing of events already resident (ebo_set_windows_device)
        t0 = time.perf_counter()
        os.environ["EBO_BUCKET"] = "host"
        ctx.set_windows(ev, offsets)
        t_host = time.perf_counter() - t0
        os.environ.pop("EBO_BUCKET")
        t0 = time.perf_counter()
        ctx.set_windows(ev, offsets)
        t_dev = time.perf_counter() - t0
        d_raw = torch.from_numpy(ev.view(np.uint8).reshape(-1, 24)).to("cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.set_windows_device(d_raw.data_ptr(), offsets)
        t_res = time.perf_counter() - t0
        extras["window_setup_mevents_per_s"] = {
            "host_bucketing_plus_upload": n_events / t_host / 1e6,
            "raw_upload_plus_device_bucketing": n_events / t_dev / 1e6,
            "device_bucketing_resident_events": n_events / t_res / 1e6}
        del d_raw
        # the reference's own default objective (edge / structure-tensor loss), value+Jacobian
        We = min(Wn, 64)
        ce = ebo.Context(device=local, image_w=cfg["image"][0], image_h=cfg["image"][1],
                         patch_w=cfg["patch"][0], patch_h=cfg["patch"][1], loss=ebo.LOSS_EDGE,
                         tv_weight=0.0, max_events=int(offsets[We]), max_windows=We)
        ce.set_stream(stream.cuda_stream)
        ce.set_windows(ev[: int(offsets[We])], offsets[: We + 1])
        fe = d_flows[:We].contiguous()
        oe = torch.zeros((We * P, 3), dtype=torch.float64, device="cuda")
        for _ in range(2):
            ce.eval_device(fe.data_ptr(), 1, oe.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(10):
            ce.eval_device(fe.data_ptr(), 1, oe.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ems = e0.elapsed_time(e1) / 10
        extras["edge_loss_value_jacobian"] = {"ms": ems, "windows": We,
                                              "mevents_per_s": int(offsets[We]) / (ems * 1e-3) / 1e6}
        ce.close()
        # the 1280x720 stream of BASELINE configs[3] (1024 patches, 1 M events per window) on this
        # one GPU: same kernel, 8 windows per launch
        c4 = synth.CONFIGS[4]
        ev4, off4, gt4 = synth.make_stream(4, 8)
        cc = ebo.Context(device=local, image_w=c4["image"][0], image_h=c4["image"][1], patch_w=c4["patch"][0],
                         patch_h=c4["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev4),
                         max_windows=8)
        cc.set_stream(stream.cuda_stream)
        cc.set_windows(ev4, off4)
        f4 = torch.from_numpy(gt4 * 0.5).to("cuda")
        o4 = torch.zeros((8 * cc.P, 3), dtype=torch.float64, device="cuda")
        for _ in range(2):
            cc.eval_device(f4.data_ptr(), 1, o4.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(10):
            cc.eval_device(f4.data_ptr(), 1, o4.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms4 = e0.elapsed_time(e1) / 10
        extras["c4_1280x720_value_jacobian"] = {"ms": ms4, "windows": 8, "patches_per_window": cc.P,
                                                "mevents_per_s": len(ev4) / (ms4 * 1e-3) / 1e6}
        cc.close()
        del ev4, f4, o4
        # the reference's own call (240x180, 20x20 patches, 15 k events, edge loss, TV-coupled
        # global LM: FeatureDetector::compensateEventsContrast as shipped), 64 windows in lock step
        rcfg = dict(name="reference default", image=(240, 180), patch=(20, 20), events=15000, index=0)
        rev, roff, _ = synth.make_stream(rcfg, 64)
        cr = ebo.Context(device=local, image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE,
                         max_events=len(rev), max_windows=64)
        cr.set_windows(rev, roff)
        cr.solve(ebo.default_solver())
        t0 = time.perf_counter()
        _, rs = cr.solve(ebo.default_solver())
        t_ref = time.perf_counter() - t0
        extras["reference_default_call"] = {"windows": 64, "ms_per_window": t_ref * 1e3 / 64,
                                            "iterations": rs[0].iterations}
        cr.close()
        # per-feature tracker objective (Optimizer::optimize's solve), 100 tracked 25x25 patches
        rng = np.random.default_rng(7)
        ys, xs = np.mgrid[0:180, 0:240].astype(np.float64)
        img = sum(rng.uniform(-1, 1) * np.exp(-((xs - rng.uniform(0, 240)) ** 2 + (ys - rng.uniform(0, 180)) ** 2)
                                              / (2 * rng.uniform(3, 9) ** 2)) for _ in range(40))
        gx, gy = np.zeros_like(img), np.zeros_like(img)
        gx[:, 1:-1] = 0.5 * (img[:, 2:] - img[:, :-2])
        gy[1:-1, :] = 0.5 * (img[2:, :] - img[:-2, :])
        co = ebo.Context(device=local, image_w=240, image_h=180)
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
        base = cpu_baseline(synth, args.config, args.cpu_seconds) if world == 1 else None
        # HBM traffic of the dominant kernel per launch, from the committed PMC profile of
        # this same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            if t.get("workload") == cfg["name"] and t.get("windows") == Wn:
                traffic = t.get("eval_kernel_hbm_bytes_per_launch")
        value = total_events * args.steps / dt / 1e6
        line = {
            "metric": "Mevents/s warped+scored (value+Jacobian of the variance-contrast objective)",
            "value": value, "unit": "Mevents/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"], "windows_per_gpu_per_step": Wn,
                       "events_per_gpu_per_step": n_events, "scored_events_per_gpu_per_step": active_events,
                       "patches_per_window": P, "loss": "variance", "grad": "jet",
                       "parallelism": "windows sharded over %d GPU(s)%s" % (
                           world, ", RCCL all-gather of (r,J) per step" if exchange
                           else (", no data-path collective" if world > 1 else ""))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_eval3<true>", "kernel_ms": kern_ms,
                         "note": "algorithmic 8 B/event-evaluation; the kernel is bound by LDS "
                                 "atomics + f64 VALU, not HBM (DESIGN.md section 4)"},
            "cpu_baseline": base,
            "extras": extras,
        }
        print(json.dumps(line))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
