"""One-off sweep on the GPU box: the device-resident per-patch solves with and without the reuse of what a cost
evaluation left in LDS (EBO_SOLVE_NO_REUSE=1) on the random windows of tests/test_gpu_random.py -- odd sensor and
patch sizes, sparse to dense windows, events on and beyond the border, penalty branches, images that need sub-bands.
Flows and statistics must be equal bit for bit.  usage: diag_reuse_campaign.py <first> <last> [ENV_VAR [global]]
(ENV_VAR: the knob whose "1" / "0" setting is compared with its absence, default EBO_SOLVE_NO_REUSE; "global": the
TV-coupled lock-step solve instead of the per-patch device solve, e.g. with EBO_SOLVE_SPECULATE, value 0)"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: F401,E402  (its HIP runtime first, see tests/conftest.py)
# the switches compared here exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(HERE), "event-based-odomety_amd", "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
import test_gpu_random as T  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
KNOB = sys.argv[3] if len(sys.argv) > 3 else "EBO_SOLVE_NO_REUSE"
GLOBAL = len(sys.argv) > 4 and sys.argv[4] == "global"
OFF = "0" if KNOB == "EBO_SOLVE_SPECULATE" else "1"
bad = []
solved = 0
for seed in range(first, last):
    cs = T.random_case(seed)
    ev = ebo.make_events(cs["x"], cs["y"], cs["t"], cs["sign"])
    for loss in (ebo.LOSS_VARIANCE, ebo.LOSS_EDGE):
        out = []
        for no_reuse in (False, True):
            os.environ.pop(KNOB, None)
            if no_reuse:
                os.environ[KNOB] = OFF
            with ebo.Context(image_w=cs["w"], image_h=cs["h"], patch_w=cs["pw"], patch_h=cs["ph"], loss=loss,
                             tv_weight=1e3 if GLOBAL else 0.0, min_events=3 + seed % 40, max_events=cs["n"]) as c:
                c.set_window(ev)
                opts = ebo.default_solver(mode=ebo.SOLVE_GLOBAL if GLOBAL else ebo.SOLVE_INDEPENDENT)
                opts.max_num_iterations = 5 + seed % 30
                flows, summ = c.solve(opts)
                out.append((flows.copy(), [(s.iterations, s.termination, s.num_evals_cost, s.num_evals_jac) for s in summ]))
        os.environ.pop(KNOB, None)
        solved += 1
        if not (np.array_equal(out[0][0], out[1][0], equal_nan=True) and out[0][1] == out[1][1]):
            bad.append((seed, loss))
            print("seed %d loss %d DIFFERS: max |d| %.3e" % (seed, loss, np.nanmax(np.abs(out[0][0] - out[1][0]))), flush=True)
    if seed % 50 == 0:
        print("... seed %d done, %d differences so far" % (seed, len(bad)), flush=True)
print("seeds [%d, %d): %d solves compared, %d differences %s" % (first, last, solved, len(bad), bad))
