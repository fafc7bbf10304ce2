"""Diagnostic (GPU box): how fast do the device solver and the oracle solver drift
apart as the iteration cap grows?  Prints max |flow_gpu - flow_oracle| per cap, next
to the oracle's own sensitivity to a 1-ulp change of compensateScale."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")

for config, n in ((0, 15000), (2, 50000)):
    cfg = synth.CONFIGS[config]
    ev, _ = synth.make_window(config, n_events=n)
    kw = dict(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1])
    prm = orc.default_params(loss=1, tv_weight=0.0, **kw)
    prm2 = orc.default_params(loss=1, tv_weight=0.0, scale=np.nextafter(1e-3, 1), **kw)
    with ebo.Context(loss=ebo.LOSS_VARIANCE, tv_weight=0.0, **kw) as c:
        c.set_window(ev)
        for iters in (1, 2, 4, 8, 10, 12, 14, 16, 20, 25, 30, 40, 50):
            fg, sg = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=iters)
            fo, _, so = orc.compensate_events_contrast(ev, prm, orc.default_solver(mode=1, max_num_iterations=iters), want_image=False)
            f2, _, _ = orc.compensate_events_contrast(ev, prm2, orc.default_solver(mode=1, max_num_iterations=iters), want_image=False)
            d = np.abs(fg[0] - fo).max(axis=1)
            print("config %d iters %2d  gpu-vs-oracle max %.3e median %.3e  #patches>1e-5: %3d | oracle 1-ulp sens %.3e | evals gpu %d/%d oracle %d/%d"
                  % (config, iters, d.max(), np.median(d[d > 0]) if (d > 0).any() else 0.0, int((d > 1e-5).sum()),
                     np.abs(fo - f2).max(), sg[0].num_evals_cost, sg[0].num_evals_jac, so.num_evals_cost, so.num_evals_jac))
