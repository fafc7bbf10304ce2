"""Window set-up (upload + device bucketing) rates for the input forms of the boundary.
usage: time_ingest.py [config] [windows]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")

config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Wn = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synth.CONFIGS[config]
ev, offsets, gt = synth.make_stream(config, Wn)
n = len(ev)
ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                  loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=n, max_windows=Wn)
t_base = np.array([int(ev["t_us"][int(offsets[w])]) for w in range(Wn)], dtype=np.int64)
ev8 = np.concatenate([ebo.pack_events8(ev[int(offsets[w]):int(offsets[w + 1])], t_base[w]) for w in range(Wn)])
pin8 = torch.from_numpy(ev8.view(np.uint8).reshape(-1, 8)).pin_memory()
pin24 = torch.from_numpy(ev.view(np.uint8).reshape(-1, 24)).pin_memory()
d8 = pin8.to("cuda")
d24 = pin24.to("cuda")
ev24p = pin24.numpy().view(ebo.EVENT_DTYPE).reshape(-1)


def best(fn, reps=5):
    fn()
    b = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        b = dt if b is None else min(b, dt)
    return b


rows = [
    ("24 B pageable host -> device bucketing", lambda: ctx.set_windows(ev, offsets)),
    ("24 B pinned host   -> device bucketing", lambda: ctx.set_windows(ev24p, offsets)),
    ("8 B pageable host  -> device bucketing", lambda: ctx.set_windows8(ev8, t_base, offsets)),
    ("8 B pinned host    -> device bucketing", lambda: ctx.set_windows8(pin8.data_ptr(), t_base, offsets)),
    ("24 B resident      -> device bucketing", lambda: ctx.set_windows_device(d24.data_ptr(), offsets)),
    ("8 B resident       -> device bucketing", lambda: ctx.set_windows8(d8.data_ptr(), t_base, offsets, device=True)),
]
def copy8():
    d8.copy_(pin8, non_blocking=True)
    torch.cuda.synchronize()


def copy8_pageable():
    d8.copy_(torch.from_numpy(ev8.view(np.uint8).reshape(-1, 8)))
    torch.cuda.synchronize()


rows += [("(plain copy of the 8 B records, pinned)", copy8), ("(plain copy of the 8 B records, pageable)", copy8_pageable)]
for name, fn in rows:
    t = best(fn)
    print("cfg %d x %d windows (%d events): %-40s %7.3f ms  %8.0f Mev/s" % (config, Wn, n, name, t * 1e3, n / t / 1e6), flush=True)
ctx.close()
