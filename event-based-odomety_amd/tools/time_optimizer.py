"""Times ebo_optimizer_solve (the ceres::Solve of tracker::Optimizer::optimize, batched over
tracked patches) and ebo_optimizer_eval on the device.  usage: time_optimizer.py [N_PATCHES ...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ebo = importlib.import_module("event-based-odomety_amd")


def scene(w, h, seed):
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w))
    for _ in range(40):
        cx, cy, s = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, 9)
        img += rng.uniform(-1, 1) * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * s * s))
    gx, gy = np.zeros_like(img), np.zeros_like(img)
    gx[:, 1:-1] = 0.5 * (img[:, 2:] - img[:, :-2])
    gy[1:-1, :] = 0.5 * (img[2:, :] - img[:-2, :])
    return gx, gy


def main():
    counts = [int(a) for a in sys.argv[1:]] or [1, 100, 1000]
    w, h = 240, 180
    gx, gy = scene(w, h, 1)
    p = ebo.default_params()
    p.image_w, p.image_h = w, h
    c = ebo.Context(p)
    c.optimizer_set_grad(gx, gy)
    rng = np.random.default_rng(2)
    for n in counts:
        rects = np.stack([rng.uniform(5, w - 30, n), rng.uniform(5, h - 30, n), np.full(n, 25.0), np.full(n, 25.0)], 1)
        nablas = [rng.integers(-3, 4, (25, 25)).astype(np.float64) for _ in range(n)]
        poses = np.tile([1.0, 0.0, 0.0, 0.0], (n, 1))
        fds = rng.uniform(0, 6.28, n)
        best_s = best_e = None
        for _ in range(5):
            t0 = time.perf_counter()
            po, fo, sums = c.optimizer_solve(rects, nablas, poses, fds, normalize=True)
            t1 = time.perf_counter()
            c.optimizer_eval(rects, [a / np.linalg.norm(a) for a in nablas], poses, fds)
            t2 = time.perf_counter()
            best_s = t1 - t0 if best_s is None else min(best_s, t1 - t0)
            best_e = t2 - t1 if best_e is None else min(best_e, t2 - t1)
        ev = sum(s.num_evals_cost + s.num_evals_jac for s in sums)
        print("%5d patches of 25x25: solve %.3f ms (%.1f us/patch, %d LM iterations max, %d evaluations, "
              "%.1f M residual-evals/s) | value+Jacobian evaluation %.3f ms (host call incl. copies)"
              % (n, best_s * 1e3, best_s * 1e6 / n, max(s.iterations for s in sums), ev,
                 ev * 625 / best_s / 1e6, best_e * 1e3), flush=True)
    c.close()


if __name__ == "__main__":
    main()
