"""Experiment (round 3): does a STATIC bank-aware order of a unit's events pay?  The events of every unit are
re-ordered once on the host -- aligned groups of 16 consecutive events get different residues of
(y * pitch + x) mod 16, the LDS bank pair of their footprint origin at zero flow -- keeping each unit's first and
last event in place (reference time).  Any order inside a unit is allowed: the accumulation is exact, so the
results must be bit-identical.  usage: static_deal.py <config> <windows>"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")


def deal(ev, cfg):
    iw, ih = cfg["image"]
    pw, ph = cfg["patch"]
    npx, npy = iw // pw, ih // ph
    px = np.minimum(ev["x"] // pw, npx - 1)
    py = np.minimum(ev["y"] // ph, npy - 1)
    unit = py * npx + px
    out = ev.copy()
    order = np.argsort(unit, kind="stable")
    bounds = np.searchsorted(unit[order], np.arange(npx * npy + 1))
    for u in range(npx * npy):
        idx = order[bounds[u]:bounds[u + 1]]  # positions of the unit's events in the window, in time order
        if len(idx) < 34:
            continue
        x, y = ev["x"][idx], ev["y"][idx]
        pitch = (int(x.max() - x.min()) + 7) | 1
        key = (y * pitch + x) % 16
        inner = np.arange(1, len(idx) - 1)
        buckets = [list(inner[key[inner] == r]) for r in range(16)]
        seq = []
        # lane slot of position p in the unit is p mod 16 within its aligned group: position 0 is the first event
        # (kept), so the group's other 15 slots are filled with 15 different residues != its own where possible
        while any(buckets):
            for r in np.argsort([-len(b) for b in buckets]):
                if buckets[r]:
                    seq.append(buckets[r].pop())
        perm = np.concatenate([[0], np.array(seq, dtype=np.int64), [len(idx) - 1]])
        out[idx] = ev[idx[perm]]
    return out


def main():
    config, windows = int(sys.argv[1]), int(sys.argv[2])
    cfg = synth.CONFIGS[config]
    evs, gts = [], []
    for w in range(windows):
        e, g = synth.make_window(config, window=w)
        evs.append(e)
        gts.append(g)
    gt = np.stack(gts)
    offsets = np.zeros(windows + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = {}
    for name, evl in (("time order", evs), ("dealt", [deal(e, cfg) for e in evs])):
        ev = np.concatenate(evl)
        ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                          patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0,
                          max_events=len(ev), max_windows=windows)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_windows(ev, offsets)
        d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
        for scale in (0.0, 0.5, 1.0):
            d_flows = torch.from_numpy(gt * scale).to("cuda")
            res = []
            for jac in (1, 0):
                for _ in range(3):
                    ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
                torch.cuda.synchronize()
                e0.record(stream)
                for _ in range(20):
                    ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
                e1.record(stream)
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 20)
                if jac:
                    out = d_out.cpu().numpy().copy()
            r = ref.setdefault(scale, out)
            print("cfg %d win %d flows %.1f*gt [%-10s] jac %7.3f ms | val %7.3f ms | max rel diff %.1e"
                  % (config, windows, scale, name, res[0], res[1], np.abs(out - r).max() / np.abs(r).max()), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
