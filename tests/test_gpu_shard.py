"""A SINGLE window sharded over ranks by rows of the patch grid (C4 = 128 patches per
GPU): rank r takes the patches of its rows and the events inside them (ebo_shard_range +
ebo_set_patches); evaluations and per-patch solves of the shards, concatenated in rank order (what
the all-gather delivers), equal the whole window's (values bit for bit: a patch's result depends on its own events only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,world", [(0, 2), (2, 3), (3, 8)])
def test_patch_row_shards_equal_the_whole_window(ebo, synth, config, world):
    cfg = synth.CONFIGS[config]
    ev, gt = synth.make_window(config, n_events=min(cfg["events"], 60000))
    kw = dict(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
              loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev))
    flows = gt * 0.5
    with ebo.Context(**kw) as c:
        c.set_window(ev)
        npx, npy, P = c.npx, c.npy, c.P
        r, J = c.eval(flows)
        solved, _ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=8)
        rects = [c.patch_rect(p % npx, p // npx) for p in range(P)]
        active = [c.patch_info(p)[1] for p in range(P)]
    parts_r, parts_J, parts_s = [], [], []
    for rank in range(world):
        b, e = ebo.shard_range(npy, rank, world)  # rows of the patch grid
        mine = list(range(b * npx, e * npx))
        evs, offs = [], [0]
        for p in mine:
            x0, y0, pw, ph = rects[p]
            sel = (ev["x"] >= x0) & (ev["x"] < x0 + pw) & (ev["y"] >= y0) & (ev["y"] < y0 + ph)
            evs.append(ev[sel])
            offs.append(offs[-1] + int(sel.sum()))
        if not mine:
            continue
        with ebo.Context(**kw) as c:
            c.set_patches(np.concatenate(evs), offs, [rects[p] for p in mine])
            rr, JJ = c.eval(flows[mine])
            ss, _ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=8)
        parts_r.append(rr[0])
        parts_J.append(JJ[0])
        parts_s.append(ss[0])
    rr, JJ, ss = np.concatenate(parts_r), np.concatenate(parts_J), np.concatenate(parts_s)
    act = np.array(active)
    # the reference time of a shard's patch is its own events' (contrast_functor.h:18-20), the image
    # is accumulated exactly: the value is identical; the Jacobian's f64 partial sums may be tiled
    # differently (the row tiling follows the largest rect a context holds): last bits only
    assert np.array_equal(rr[act], r[0][act])
    np.testing.assert_allclose(JJ[act], J[0][act], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ss[act], solved[0][act], rtol=0, atol=1e-9)
