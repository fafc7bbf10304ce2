"""Context lifecycle on the device: every buffer a context allocates on demand (count-image
lists and bins, evaluation staging, mode tables, edge scratch, tracker workspace, motion-field
workspace) is released by ebo_destroy, and a context that grows a buffer keeps working."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def exercise(ebo, synth, loss, n_windows):
    cfg = synth.CONFIGS[0]
    ev, offsets, gt = synth.make_stream(0, n_windows)
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                     patch_h=cfg["patch"][1], loss=loss, max_events=len(ev), max_windows=n_windows) as c:
        c.set_windows(ev, offsets)
        flows = gt * 0.5
        c.eval(flows)
        opts = ebo.default_solver()
        opts.max_num_iterations = 3
        c.solve(opts)  # lock-step host LM: pinned staging (+ mode table with several windows)
        c.count_image(ebo.COUNT_INTEGRATED)
        c.count_image(ebo.COUNT_WARPED, flows)
        field = np.zeros((n_windows, cfg["image"][1], cfg["image"][0], 2), dtype=np.float32)
        c.count_image(ebo.COUNT_FIELD, field)
        gx = np.random.RandomState(0).rand(cfg["image"][1], cfg["image"][0])
        c.optimizer_set_grad(gx, gx * 0.5)
        rects = np.array([[30.0, 40.0, 25.0, 25.0], [100.5, 60.25, 25.0, 25.0]])
        nablas = [np.random.RandomState(i).randint(-3, 4, (25, 25)).astype(np.float64) for i in range(2)]
        c.optimizer_solve(rects, nablas, np.tile([1.0, 0.0, 0.0, 0.0], (2, 1)), np.array([0.3, 1.0]), normalize=True)


def test_contexts_release_their_device_memory(ebo, synth):
    exercise(ebo, synth, ebo.LOSS_EDGE, 2)  # warm: library-level one-off allocations
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(6):
        exercise(ebo, synth, ebo.LOSS_EDGE if k % 2 else ebo.LOSS_VARIANCE, 1 + (k % 3) * 4)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


def test_growing_a_batch_reallocates_on_demand_buffers(ebo, synth, orc):
    """One context, first a small batch then the largest it allows: count images stay bit-exact
    after the lists/bins had to grow."""
    cfg = synth.CONFIGS[0]
    ev, offsets, gt = synth.make_stream(0, 6)
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                     patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev),
                     max_windows=6) as c:
        prm = orc.default_params(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                                 patch_h=cfg["patch"][1], tv_weight=0.0, loss=1)
        for n in (1, 6, 2):
            sub = ev[: offsets[n]]
            c.set_windows(sub, offsets[: n + 1])
            flows = gt[:n] * 0.7
            img = c.count_image(ebo.COUNT_WARPED, flows)
            for w in (0, n - 1):
                wev = ev[offsets[w]:offsets[w + 1]]
                assert np.array_equal(img[w], orc.final_count_image(wev, prm, flows[w]))
