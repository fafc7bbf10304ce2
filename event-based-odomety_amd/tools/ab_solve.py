"""A/B on the GPU box: device-resident per-patch solve (k_solve_independent) for env settings.
usage: ab_solve.py <config> <windows> "K=V,.." ..."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import os
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
config, windows = int(sys.argv[1]), int(sys.argv[2])
cfg = synth.CONFIGS[config]
ev, offsets, gt = synth.make_stream(config, windows)
ref = None
for st in sys.argv[3:] or [""]:
    for k in ("EBO_SOLVE_BLOCK", "EBO_LDS_KB", "EBO_SOLVE_NO_REUSE"):
        os.environ.pop(k, None)
    for kv in filter(None, st.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                     loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev), max_windows=windows) as c:
        c.set_windows(ev, offsets)
        c.solve(mode=ebo.SOLVE_INDEPENDENT)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            flows, s = c.solve(mode=ebo.SOLVE_INDEPENDENT)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    ref = flows if ref is None else ref
    print("cfg %d win %d [%-36s] solve %8.3f ms  max|dflow vs first| %.1e" % (config, windows, st, best * 1e3, np.abs(flows - ref).max()), flush=True)
