"""A/B: the bench's timed loop three ways -- one ctypes call per step, one graph launch per step, the K steps
as K graph launches inside one C call -- against the kernel's own duration by HIP events.
    python event-based-odomety_amd/tools/ab/step_loop.py [steps]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = synth.CONFIGS[3]
Wn = 64
evs, gts = zip(*[synth.make_window(3, window=w) for w in range(Wn)])
offsets = np.zeros(Wn + 1, dtype=np.uint64)
offsets[1:] = np.cumsum([len(e) for e in evs])
ev = np.concatenate(evs)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                  loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev), max_windows=Wn)
ctx.set_stream(stream.cuda_stream)
ctx.set_windows(ev, offsets)
d_flows = torch.from_numpy(np.stack(gts) * 0.5).to("cuda")
d_out = torch.zeros((Wn * ctx.P, 3), dtype=torch.float64, device="cuda")


def one():
    ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())


one()
torch.cuda.synchronize()
g = ctx.record(one)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, run in (("ctypes call per step", lambda k: [one() for _ in range(k)]),
                  ("graph launch per step", lambda k: [g.launch(1) for _ in range(k)]),
                  ("K graph launches in one call", lambda k: g.launch(k))):
    for rep in range(3):
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(stream)
        run(steps)
        e1.record(stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / steps
        print("%-30s %3d steps: wall %.4f ms/step, events %.4f ms/step" % (name, steps, wall, e0.elapsed_time(e1) / steps))
