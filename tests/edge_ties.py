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


def global_objective(r, J, active, flows, npx, npy, tv_weight, tv_huber):
    """Cost and gradient of the problem FeatureDetector::compensateEventsContrast hands to Ceres
    (feature_detector.cpp:357-396) at `flows` [P][2], from the data terms' residuals r [P] and Jacobians
    J [P][2]: 1/2 sum r_p^2 over the active patches + 1/2 sum rho(|w (x_p - x_q)|^2) over right / lower
    neighbours, rho = HuberLoss(tv_huber): s for s <= a^2, 2 a sqrt(s) - a^2 beyond (ceres/loss_function.h)."""
    act = active.astype(bool)
    cost = 0.5 * float((r[act] ** 2).sum())
    grad = np.zeros_like(flows)
    grad[act] = r[act, None] * J[act]
    x = flows.reshape(npy, npx, 2)
    g = grad.reshape(npy, npx, 2)
    a2 = tv_huber * tv_huber

    def blocks(d):
        nonlocal cost
        s = (tv_weight ** 2) * (d ** 2).sum(axis=-1)
        inl = s <= a2
        cost += 0.5 * float(np.where(inl, s, 2.0 * tv_huber * np.sqrt(np.maximum(s, 1e-300)) - a2).sum())
        rho1 = np.where(inl, 1.0, tv_huber / np.sqrt(np.maximum(s, 1e-300)))
        return (rho1 * tv_weight ** 2)[..., None] * d  # d cost / d x_p = -d cost / d x_q

    dx = blocks(x[:, :-1] - x[:, 1:])
    g[:, :-1] += dx
    g[:, 1:] -= dx
    dy = blocks(x[:-1] - x[1:])
    g[:-1] += dy
    g[1:] -= dy
    return cost, float(np.abs(grad).max())


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
            npx, npy = c.npx, c.npy
            r0, J0 = c.eval(np.zeros((len(ws), P, 2)))
            flows, summ = c.solve(ebo.default_solver())
            oracle = [orc.compensate_events_contrast(evs[k], prm, orc.default_solver(), want_image=False) for k in range(len(ws))]
            # the data terms of the HIP path at BOTH end points (its own and the oracle's), one launch each
            fo_all = np.stack([o[0] for o in oracle])
            r_hh, J_hh = c.eval(flows)
            r_ho, J_ho = c.eval(fo_all)
        for k, w in enumerate(ws):
            ro, Jo, active, _ = orc.window_eval(evs[k], prm_free, np.zeros((P, 2)))
            tie = (np.abs(J0[k] - Jo) > 1e-8 * np.abs(Jo) + 1e-7).any(axis=1) & active.astype(bool)
            value_ok = bool(np.allclose(r0[k], ro, rtol=1e-9, atol=1e-9))
            fo, _, so = oracle[k]
            # The solvers' own equivalence, whatever the trajectories: each side's objective at BOTH end points.
            # oracle objective at the HIP flows / at its own; HIP objective at the oracle's flows / at its own
            r_oh, J_oh, _, _ = orc.window_eval(evs[k], prm_free, flows[k])
            r_oo, J_oo, _, _ = orc.window_eval(evs[k], prm_free, fo)
            tvw, tvh = prm.tv_weight, prm.tv_huber
            c_oh, g_oh = global_objective(r_oh, J_oh, active, flows[k], npx, npy, tvw, tvh)
            c_oo, g_oo = global_objective(r_oo, J_oo, active, fo, npx, npy, tvw, tvh)
            c_hh, g_hh = global_objective(r_hh[k], J_hh[k], active, flows[k], npx, npy, tvw, tvh)
            c_ho, g_ho = global_objective(r_ho[k], J_ho[k], active, fo, npx, npy, tvw, tvh)
            rows.append(dict(window=w, active=int(active.sum()), tie_patches=int(tie.sum()), value_ok=value_ok,
                             cost_formula_rel=float(abs(c_oo - so.final_cost) / max(abs(so.final_cost), 1e-300)),
                             cross_cost_rel_oracle=float(abs(c_oh - c_oo) / c_oo), cross_cost_rel_hip=float(abs(c_hh - c_ho) / c_ho),
                             grad_oracle_at_hip=g_oh, grad_oracle_at_oracle=g_oo, grad_hip_at_hip=g_hh, grad_hip_at_oracle=g_ho,
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
