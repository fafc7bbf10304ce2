"""Compact A/B on the GPU box: evaluation throughput for a list of env-var settings.
usage: ab_eval.py <config> <windows> "K=V,K=V" "K=V" ...   (first setting is the reference
for the bit-difference column).  No oracle involved."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")

KEYS = ("EBO_EVAL_IMPL", "EBO_EVAL_ROT", "EBO_LDS_KB", "EBO_EVAL_TILES", "EBO_EVAL_BLOCK", "EBO_EVAL_DEAL")


def main():
    config, windows = int(sys.argv[1]), int(sys.argv[2])
    settings = sys.argv[3:] or [""]
    cfg = synth.CONFIGS[config]
    evs, gts = [], []
    for w in range(windows):
        e, g = synth.make_window(config, window=w)
        evs.append(e)
        gts.append(g)
    offsets = np.zeros(windows + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    ev = np.concatenate(evs)
    ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                      patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0,
                      max_events=len(ev), max_windows=windows)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_windows(ev, offsets)
    gt = np.stack(gts)
    d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    refs = {}
    for scale in (0.0, 0.5, 1.0):
        d_flows = torch.from_numpy(gt * scale).to("cuda")
        for st in settings:
            for k in KEYS:
                os.environ.pop(k, None)
            for kv in filter(None, st.split(",")):
                k, v = kv.split("=")
                os.environ[k] = v
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
            ref = refs.setdefault(scale, out)
            diff = np.abs(out - ref).max() / max(np.abs(ref).max(), 1e-300)
            print("cfg %d win %d flows %.1f*gt [%-52s] jac %7.3f ms %8.0f Mev/s | val %7.3f ms %8.0f Mev/s | d=%.0e"
                  % (config, windows, scale, st, res[0], len(ev) / res[0] / 1e3, res[1], len(ev) / res[1] / 1e3, diff),
                  flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
