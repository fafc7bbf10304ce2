"""Shared by tests/test_gpu_edge_ties.py and tools: the reference-default call
(FeatureDetector::compensateEventsContrast as shipped: 240x180, 20x20 patches, 15 k events, edge
loss, TV-coupled global LM, 50 iterations) on many seeded windows, HIP path against the oracle,
with the zero-flow tie patches of every window counted.

A tie patch: at exactly zero flow -- where every solve starts (feature_detector.cpp:318-326) -- all
events sit on integer positions, symmetric pixels have mathematically equal structure-tensor
eigenvalues, and which of them is a window's argmax (hence the Jacobian, not the value) is decided
by the rounding of the image sums: one rounding per event in list order in the reference, exact
accumulation on the device.  What matters is whether the SOLVE that starts there still ends at the
same flows."""
import numpy as np


def run(ebo, orc, synth, n_windows, first_window=100, n_events=15000, batch=25):
    rows = []
    prm = orc.default_params(loss=0)
    prm_free = orc.default_params(loss=0, tv_weight=0.0)
    for b0 in range(0, n_windows, batch):
        ws = list(range(first_window + b0, first_window + min(b0 + batch, n_windows)))
        evs = [synth.make_window(0, window=w, n_events=n_events)[0] for w in ws]
        offsets = np.zeros(len(ws) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(e) for e in evs])
        ev = np.concatenate(evs)
        with ebo.Context(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE, max_events=len(ev),
                         max_windows=len(ws)) as c:
            c.set_windows(ev, offsets)
            P = c.P
            r0, J0 = c.eval(np.zeros((len(ws), P, 2)))
            flows, summ = c.solve(ebo.default_solver())
        for k, w in enumerate(ws):
            ro, Jo, active, _ = orc.window_eval(evs[k], prm_free, np.zeros((P, 2)))
            tie = (np.abs(J0[k] - Jo) > 1e-8 * np.abs(Jo) + 1e-7).any(axis=1) & active.astype(bool)
            value_ok = bool(np.allclose(r0[k], ro, rtol=1e-9, atol=1e-9))
            fo, _, so = orc.compensate_events_contrast(evs[k], prm, orc.default_solver(), want_image=False)
            rows.append(dict(window=w, active=int(active.sum()), tie_patches=int(tie.sum()), value_ok=value_ok,
                             iterations=int(summ[k].iterations), iterations_oracle=int(so.iterations),
                             termination=int(summ[k].termination), termination_oracle=int(so.termination),
                             max_dflow=float(np.abs(flows[k] - fo).max()),
                             final_cost_rel=float(abs(summ[k].final_cost - so.final_cost) / max(abs(so.final_cost), 1e-300))))
    return rows


def table(rows):
    lines = ["| window | active patches | zero-flow tie patches | iterations (HIP / oracle) | max abs(flow - oracle) | rel. final cost |",
             "|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %d | %d | %d | %d / %d | %.2e | %.1e |" % (r["window"], r["active"], r["tie_patches"], r["iterations"],
                                                                 r["iterations_oracle"], r["max_dflow"], r["final_cost_rel"]))
    return "\n".join(lines)
