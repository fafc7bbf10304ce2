import sys, os, importlib, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import orc, edge_ties
import test_gpu_random as tr
ebo = importlib.import_module("event-based-odomety_amd")
tot = dict(active=0, masked=0, mismatch=0, mismatch_unmasked=0)
for eps in (1, 4, 8):
    tot = dict(active=0, masked=0, mismatch=0, mismatch_unmasked=0)
    for seed in list(range(24)) + list(range(200, 230)):
        cs = tr.random_case(seed)
        ev = ebo.make_events(cs["x"], cs["y"], cs["t"], cs["sign"])
        with ebo.Context(image_w=cs["w"], image_h=cs["h"], patch_w=cs["pw"], patch_h=cs["ph"], loss=ebo.LOSS_EDGE, tv_weight=0.0,
                         min_events=3, max_events=cs["n"]) as c:
            c.set_window(ev)
            p = c.params
            prm = orc.default_params(image_w=p.image_w, image_h=p.image_h, patch_w=p.patch_w, patch_h=p.patch_h, tv_weight=0.0, min_events=3, loss=0)
            r, J = c.eval(np.zeros((c.P, 2)))
            mask, J0, active = edge_ties.zero_flow_tie_mask(orc, ev, prm, c.P, n_perm=eps)
            mism = (np.abs(J[0] - J0) > 1e-8 * np.abs(J0) + 1e-7).any(axis=1) & active
            tot["active"] += int(active.sum()); tot["masked"] += int(mask.sum()); tot["mismatch"] += int(mism.sum())
            tot["mismatch_unmasked"] += int((mism & ~mask).sum())
    print("eps", eps, tot, flush=True)
