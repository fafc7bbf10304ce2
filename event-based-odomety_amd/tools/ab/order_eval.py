"""A/B: does the order of a unit's events matter to k_eval3 (LDS atomics of neighbouring lanes)?  The canonical order
is dt-major (positions of consecutive events are uncorrelated); a y-major / row-class order would let the warped
count image read sub-ranges of a unit, but puts consecutive events on neighbouring pixels.
    python event-based-odomety_amd/tools/ab/order_eval.py [config] [windows]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("EBO_LIB_PATH", os.path.join(ROOT, "event-based-odomety_amd", "libebo_hip_ab.so"))
os.environ["EBO_KEEP_ORDER"] = "1"
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Wn = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synth.CONFIGS[ci]
iw, ih = cfg["image"]
pw, ph = cfg["patch"]
npx, npy, rects = synth.grid_rects(cfg["image"], cfg["patch"])
P = npx * npy
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
LOSS = ebo.LOSS_EDGE if os.environ.get("AB_LOSS") == "edge" else ebo.LOSS_VARIANCE
ORDERS = tuple(os.environ.get("AB_ORDERS", "dt-major (canonical),stride 37,stride 101").split(","))
for order in ORDERS:
    evs, offs, allrects, flows = [], [0], [], []
    for w in range(Wn):
        ev, gt = synth.make_window(ci, window=w)
        gx = np.minimum(ev["x"] // pw, npx - 1)
        gy = np.minimum(ev["y"] // ph, npy - 1)
        pid = gy * npx + gx
        for p in range(P):
            sel = ev[pid == p]
            if len(sel):
                tref = (int(sel["t_us"][0]) + int(sel["t_us"][-1])) // 2
                dt = (tref - sel["t_us"]).astype(np.int64)
                if order.startswith("dt"):
                    key = np.argsort(dt.astype(np.uint32), kind="stable")
                elif order.startswith("row"):
                    cls = ((sel["y"] - rects[p][1]) * 8) // rects[p][3]
                    key = np.lexsort((dt.astype(np.uint32), cls))
                elif order.startswith("y-major"):
                    key = np.lexsort((sel["x"], sel["y"]))
                else:
                    base = np.argsort(dt.astype(np.uint32), kind="stable")
                    n = len(sel)
                    if order.startswith("permuted"):
                        key = base[np.random.default_rng(p + 1000 * w).permutation(n)]
                    elif order.startswith("spatial"):
                        # ranks by position (row-major, column-major or Morton), then dealt out with stride 127
                        kind = order.split()[1]
                        xs, ys = sel["x"].astype(np.int64) - rects[p][0], sel["y"].astype(np.int64) - rects[p][1]
                        if kind == "rows":
                            base2 = np.lexsort((dt.astype(np.uint32), xs, ys))
                        elif kind == "cols":
                            base2 = np.lexsort((dt.astype(np.uint32), ys, xs))
                        else:
                            def spread(v):
                                v = v & 0xFF
                                v = (v | (v << 4)) & 0x0F0F
                                v = (v | (v << 2)) & 0x3333
                                v = (v | (v << 1)) & 0x5555
                                return v
                            base2 = np.lexsort((dt.astype(np.uint32), spread(xs) | (spread(ys) << 1)))
                        st = int(order.split()[2]) if len(order.split()) > 2 else 127
                        st = max(st % max(n, 1), 1)
                        while np.gcd(st, n) != 1:
                            st += 1
                        key = base2[(np.arange(n) * st) % n]
                    elif order.startswith("stride"):
                        what = order.split()[1]
                        st = {"golden": int(round(n * 0.6180339887)), "n/64+1": n // 64 + 1, "n/128+1": n // 128 + 1,
                              "sqrt": int(round(np.sqrt(n)))}.get(what) or int(what)
                        st = max(st % max(n, 1), 1)
                        while np.gcd(st, n) != 1:
                            st += 1
                        key = base[(np.arange(n) * st) % n]
                    elif order.startswith("bitrev"):
                        bits = max(int(n - 1).bit_length(), 1)
                        rev = np.array([int(format(i, "0%db" % bits)[::-1], 2) for i in range(1 << bits)])
                        key = base[rev[rev < n]]
                    else:
                        # consecutive events on different rows: sort by (rank within row, row)
                        rows = sel["y"][base]
                        rank = np.zeros(n, dtype=np.int64)
                        seen = {}
                        for i, r in enumerate(rows):
                            rank[i] = seen.get(int(r), 0)
                            seen[int(r)] = rank[i] + 1
                        key = base[np.lexsort((rows, rank))]
                # keep the unit's first / last event times: the reference time comes from them
                first, last = sel[0].copy(), sel[-1].copy()
                sel = sel[key]
            evs.append(sel)
            offs.append(offs[-1] + len(sel))
        allrects.append(rects)
        flows.append(gt * 0.5)
    ev_all = np.concatenate(evs)
    ctx = ebo.Context(image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=LOSS, tv_weight=0.0,
                      max_events=len(ev_all), max_windows=Wn)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_patches(ev_all, offs, np.concatenate(allrects))
    d_flows = torch.from_numpy(np.concatenate(flows)).to("cuda")
    d_out = torch.zeros((Wn * P, 3), dtype=torch.float64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(50):
        ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    print("%-34s %s x %d windows: %.4f ms per value+Jacobian evaluation, checksum %.6e"
          % (order, cfg["name"], Wn, e0.elapsed_time(e1) / 50, float(d_out[:, 0].sum())))
    if os.environ.get("AB_SOLVE"):
        d_sol = torch.zeros((Wn * P, 2), dtype=torch.float64, device="cuda")
        d_stats = torch.zeros((Wn * P, 4), dtype=torch.int32, device="cuda")
        opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT)
        ctx.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr())
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(3):
            ctx.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        st = d_stats.cpu().numpy()
        print("%-34s     device solve %.3f ms, evaluations %d (cost) + %d (Jacobian), flow checksum %.9e"
              % ("", e0.elapsed_time(e1) / 3, int(st[:, 1].sum()), int(st[:, 2].sum()), float(d_sol.abs().sum())))
    ctx.close()
