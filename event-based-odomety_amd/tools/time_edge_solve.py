"""Per-patch (TV-free) solve with the edge loss: device-resident LM (k_solve_edge) vs host LMs in
lock step.  usage: time_edge_solve.py [config] [windows]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")

config = int(sys.argv[1]) if len(sys.argv) > 1 else 0
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = synth.CONFIGS[config]
ev, offsets, gt = synth.make_stream(config, windows)
c = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=windows)
c.set_windows(ev, offsets)
for how in ("device", "lockstep", "device"):
    os.environ["EBO_SOLVE_EDGE"] = how
    opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        flows, ss = c.solve(opts)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("cfg %d, %d windows, %-8s: %.2f ms = %.3f ms per window; window 0: %d iterations, %d + %d evaluations"
          % (config, windows, how, best * 1e3, best * 1e3 / windows, ss[0].iterations, ss[0].num_evals_cost,
             ss[0].num_evals_jac), flush=True)
c.close()
