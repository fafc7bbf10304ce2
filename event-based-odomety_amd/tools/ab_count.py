"""Compact A/B on the GPU box: count-image kernel time for a list of env settings.
usage: ab_count.py <config> <windows> "K=V,K=V" ...   No oracle involved."""
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
if os.environ.get("AB_LIB"):  # A/B of two builds on the same box
    ebo.LIB_PATH = os.path.join(os.path.dirname(ebo.LIB_PATH), os.environ["AB_LIB"])

KEYS = ("EBO_COUNT_IMPL", "EBO_COUNT_LDS_KB", "EBO_COUNT_BLOCK", "EBO_COUNT_TILE_W", "EBO_COUNT_TILE_H")


def main():
    config, windows = int(sys.argv[1]), int(sys.argv[2])
    settings = sys.argv[3:] or [""]
    cfg = synth.CONFIGS[config]
    ev, offsets, gt = synth.make_stream(config, windows)
    ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                      patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, max_events=len(ev), max_windows=windows)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_windows(ev, offsets)
    d_flows = torch.from_numpy(gt * float(os.environ.get("AB_FLOW_SCALE", "0.8"))).to("cuda")  # AB_FLOW_SCALE=0: no unit reaches a neighbouring tile
    d_img = torch.zeros((windows, cfg["image"][1], cfg["image"][0]), dtype=torch.float64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = {}
    algo = 8 * len(ev) + d_img.numel() * 8
    for st in settings:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in filter(None, st.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
        for mode in (ebo.COUNT_INTEGRATED, ebo.COUNT_WARPED):
            aux = d_flows.data_ptr() if mode == ebo.COUNT_WARPED else 0
            # best of 5 batches of 40 launches after a warm-up of 40: single batches of 20 scattered
            # by +-5 % between identical settings (clock ramps, the copy of the previous image)
            for _ in range(40):
                ctx.count_image_device(mode, aux, d_img.data_ptr())
            torch.cuda.synchronize()
            ms = None
            for _ in range(5):
                e0.record(stream)
                for _ in range(40):
                    ctx.count_image_device(mode, aux, d_img.data_ptr())
                e1.record(stream)
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 40
                ms = t if ms is None else min(ms, t)
            img = d_img.cpu().numpy()
            same = np.array_equal(img, ref.setdefault(mode, img))
            print("cfg %d win %d mode %d [%-36s] %7.3f ms %8.0f Mev/s %7.1f GB/s algorithmic (%.1f%% of 8 TB/s) same=%s"
                  % (config, windows, mode, st, ms, len(ev) / ms / 1e3, algo / ms / 1e6, algo / ms / 1e6 / 80, same), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
