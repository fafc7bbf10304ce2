"""Diagnostic (not a test): how far apart are two correct solves of the tracker objective?
CPU: the oracle with DENSE_QR vs the oracle with the normal equations (same LM).  With --gpu:
the device solve vs both.  Shows the conditioning of the problem, not an implementation error."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orc
from test_optimizer import _batch


def main():
    gpu = "--gpu" in sys.argv
    grad, items = _batch(orc, 24)
    if gpu:
        ebo = importlib.import_module("event-based-odomety_amd")
        p = ebo.default_params()
        p.image_w, p.image_h = 240, 180
        c = ebo.Context(p)
        c.optimizer_set_grad(grad[..., 0], grad[..., 1])
    for iters in (1, 3, 5, 10, 15, 20, 30, 40):
        oq = orc.optimizer_default_solver(max_num_iterations=iters)
        on = orc.optimizer_default_solver(max_num_iterations=iters, mode=1)
        if gpu:
            od = ebo.optimizer_default_solver(max_num_iterations=iters)
            poses, fds, sums = c.optimizer_solve([it["rect"] for it in items], [it["nabla"] for it in items],
                                                 [it["start"][0] for it in items], [it["start"][1] for it in items], opts=od)
        dqn, dqd, dnd, same_it = [], [], [], 0
        for i, it in enumerate(items):
            pq, fq, sq = orc.optimizer_solve(grad, it["rect"], it["nabla"], it["start"][0], it["start"][1], opts=oq)
            pn, fn, sn = orc.optimizer_solve(grad, it["rect"], it["nabla"], it["start"][0], it["start"][1], opts=on)
            dqn.append(max(np.abs(pq - pn).max(), abs(fq - fn)))
            same_it += int(sq.iterations == sn.iterations)
            if gpu:
                dqd.append(max(np.abs(pq - poses[i]).max(), abs(fq - fds[i])))
                dnd.append(max(np.abs(pn - poses[i]).max(), abs(fn - fds[i])))
        line = "iters %2d: |QR - normal eq| max %.1e median %.1e (same iteration count %d/%d)" % (
            iters, max(dqn), float(np.median(dqn)), same_it, len(items))
        if gpu:
            line += " | device vs QR max %.1e median %.1e | device vs normal eq max %.1e" % (
                max(dqd), float(np.median(dqd)), max(dnd))
        print(line, flush=True)


if __name__ == "__main__":
    main()
