"""Latency of one small window's set-up (ebo_set_windows on 15 k host events: upload + device
bucketing + unit table back) with the stage split of EBO_INGEST_TRACE.  usage: time_set_window.py [N]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import os
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
ev, _ = synth.make_window(0, n_events=n)
with ebo.Context(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE, max_events=n) as c:
    for _ in range(5):
        c.set_window(ev)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); c.set_window(ev); ts.append(time.perf_counter() - t0)
    print("set_window of %d events: best %.3f ms, median %.3f ms" % (n, min(ts) * 1e3, sorted(ts)[10] * 1e3))
    os.environ["EBO_INGEST_TRACE"] = "1"
    c.set_window(ev)
