"""Diagnostic (GPU box): wall time of ONE FeatureDetector::compensateEventsContrast with every
reference default (edge loss, TV 1e3/Huber 10, one global LM, <= 50 iterations) on a 15 000-event
window, HIP path vs CPU oracle (single thread, like the reference)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
ev, _ = synth.make_window(0, n_events=15000)
for loss in (ebo.LOSS_EDGE, ebo.LOSS_VARIANCE):
    with ebo.Context(loss=loss) as c:
        c.compensate_events_contrast(ev)  # warm-up (allocations, code load)
        t0 = time.perf_counter()
        for _ in range(5):
            flows, img, s = c.compensate_events_contrast(ev)
        tg = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    fo, io, so = orc.compensate_events_contrast(ev, orc.default_params(loss=loss), orc.default_solver())
    tc = time.perf_counter() - t0
    print("loss %s: GPU %.2f ms per call (iterations %d, evals %d+%d) | CPU oracle %.2f s | speed-up %.0fx | max|dflow| %.2e"
          % ("edge" if loss == 0 else "variance", tg * 1e3, s.iterations, s.num_evals_cost // 97, s.num_evals_jac // 97,
             tc, tc / tg, np.abs(flows - fo).max()))
