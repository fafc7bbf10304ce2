"""Tuning sweep on the GPU box: objective-evaluation throughput vs row tiles and
workgroup size (EBO_EVAL_TILES / EBO_EVAL_BLOCK).  No oracle involved."""
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


def run(config=2, windows=128, steps=10, jac=1, tiles_list=(1, 2, 3, 4, 6), blocks=(256, 512, 1024)):
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
    d_flows = torch.from_numpy(np.stack(gts) * 0.5).to("cuda")
    d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for tiles in tiles_list:
        for block in blocks:
            os.environ["EBO_EVAL_TILES"] = str(tiles)
            os.environ["EBO_EVAL_BLOCK"] = str(block)
            try:
                for _ in range(2):
                    ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
                torch.cuda.synchronize()
                e0.record(stream)
                for _ in range(steps):
                    ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
                e1.record(stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / steps
                print("config %d windows %d jac %d tiles %d block %4d : %8.3f ms  %9.1f Mev/s"
                      % (config, windows, jac, tiles, block, ms, len(ev) / ms / 1e3), flush=True)
            except ebo.EboError as e:
                print("tiles %d block %d: %s" % (tiles, block, e))
    ctx.close()


if __name__ == "__main__":
    cfgs = [int(a) for a in sys.argv[1:]] or [2]
    for c in cfgs:
        run(c, jac=1)
        run(c, jac=0, tiles_list=(1, 2, 3))
