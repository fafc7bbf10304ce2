"""Times FeatureDetector::compensateEventsContrast as ONE C-ABI call on host events
(ebo_compensate_events_contrast: upload + bucketing + solve + final count image), split into its
stages, for the reference-default configuration.  usage: time_one_call.py [N_EVENTS]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")


def best(f, n=7):
    b = None
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        dt = time.perf_counter() - t0
        b = dt if b is None else min(b, dt)
    return b * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
    ev, _ = synth.make_window(0, n_events=n)
    for loss in (ebo.LOSS_EDGE, ebo.LOSS_VARIANCE):
        with ebo.Context(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=loss, max_events=n) as c:
            opts = ebo.default_solver()
            c.compensate_events_contrast(ev, opts)
            one = best(lambda: c.compensate_events_contrast(ev, opts))
            setw = best(lambda: c.set_window(ev))
            flows, _ = c.solve(opts)
            solve = best(lambda: c.solve(opts))
            img = best(lambda: c.count_image(ebo.COUNT_WARPED, flows[0]))
            integ = best(lambda: c.count_image(ebo.COUNT_INTEGRATED))
            print("loss %d, %d events: one call %.3f ms = set_window %.3f + solve %.3f + final image %.3f "
                  "(integrateEvents %.3f)" % (loss, n, one, setw, solve, img, integ), flush=True)


if __name__ == "__main__":
    main()
