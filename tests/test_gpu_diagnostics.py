"""The diagnostic entry points bench.py builds its rooflines on (include/ebo.h: ebo_edge_work_stats, ebo_lds_rates,
ebo_stream_yardstick_device): what they count is what the launch did."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_edge_work_stats_count_the_launch(ebo, synth):
    import torch
    cfg = synth.CONFIGS[0]
    ev, offsets, gt = synth.make_stream(0, 6, n_events=9000)
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                     loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=6) as c:
        c.set_windows(ev, offsets)
        info = [[c.patch_info(p, w) for p in range(c.P)] for w in range(6)]
        d_flows = torch.from_numpy(gt * 0.5).to("cuda")
        r, J = c.eval(gt * 0.5)
        st = c.edge_work_stats(d_flows.data_ptr(), True)
        sv = c.edge_work_stats(d_flows.data_ptr(), False)
        r2, J2 = c.eval(gt * 0.5)
    active = [(n, a) for w in info for (n, a, _) in w if a]
    # every active unit of these windows has events inside its canvas: all pass the empty-window test
    assert st["units"] == len(active) and st["events"] == sum(n for n, _ in active)
    canvas = 9 * cfg["patch"][0] * cfg["patch"][1]
    assert 0 < st["eigen_pixels"] < st["box_pixels"] <= st["units"] * canvas
    assert 0 < st["argmax_entries"] <= st["nms_windows"] < st["box_pixels"] // 4 + st["units"]
    assert sv["argmax_entries"] == 0
    assert {k: sv[k] for k in sv if k != "argmax_entries"} == {k: st[k] for k in st if k != "argmax_entries"}
    # counting changes nothing
    assert np.array_equal(r, r2) and np.array_equal(J, J2)
    with ebo.Context(loss=ebo.LOSS_VARIANCE, max_events=100) as c:
        c.set_window(ev[:100])
        with pytest.raises(ebo.EboError) as err:
            c.edge_work_stats(d_flows.data_ptr(), True)
        assert err.value.code == ebo.ERR_STATE


def test_lds_rates_and_stream_yardstick(ebo, synth):
    import torch
    cfg = synth.CONFIGS[2]
    ev, offsets, _ = synth.make_stream(2, 8, n_events=20000)
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                     max_events=len(ev), max_windows=8) as c:
        atomics, reads = c.lds_rates()
        # a CU does several 64-bit LDS operations per clock: thousands of G operations per second chip-wide
        assert 500.0 < atomics < 20000.0 and 500.0 < reads < 40000.0
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_windows(ev, offsets)
        d_img = torch.ones((8, cfg["image"][1], cfg["image"][0]), dtype=torch.float64, device="cuda")
        moved = c.stream_yardstick_device(d_img.data_ptr())
        torch.cuda.synchronize()
        assert moved == 8 * (len(ev) // 2) * 2 + d_img.numel() * 8
        assert float(d_img.abs().sum()) == 0.0  # every pixel written (with 0.0)
        img = c.count_image(ebo.COUNT_INTEGRATED)  # and the events are untouched
        assert img.sum() == len(ev)
