"""ebo_graph_*: a recorded step replays with the same bits; a recording always ends."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(ebo, synth, loss):
    ev, offsets, gt = synth.make_stream(0, 3)
    c = ebo.Context(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=loss, tv_weight=0.0,
                    max_events=len(ev), max_windows=3)
    c.set_windows(ev, offsets)
    return c, ev, offsets, gt


@pytest.mark.parametrize("loss", ["variance", "edge"])
def test_recorded_evaluation_replays_bit_for_bit(ebo, synth, loss):
    import torch
    L = ebo.LOSS_VARIANCE if loss == "variance" else ebo.LOSS_EDGE
    c, ev, offsets, gt = _ctx(ebo, synth, L)
    with c:
        stream = torch.cuda.Stream()  # a created stream: the default stream cannot be recorded
        c.set_stream(stream.cuda_stream)
        torch.cuda.set_stream(stream)
        d_flows = torch.from_numpy(gt * 0.5).to("cuda")
        d_ref = torch.zeros((3 * c.P, 3), dtype=torch.float64, device="cuda")
        d_out = torch.zeros_like(d_ref)
        c.eval_device(d_flows.data_ptr(), 1, d_ref.data_ptr())  # also allocates the edge work table
        torch.cuda.synchronize()
        g = c.record(lambda: c.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr()))
        assert float(d_out.abs().sum()) == 0.0  # recording runs nothing
        g.launch(3)
        torch.cuda.synchronize()
        assert torch.equal(d_out, d_ref)
        # the graph reads the flows it was recorded with: new values in the same buffer, same graph
        d_flows.mul_(0.5)
        c.eval_device(d_flows.data_ptr(), 1, d_ref.data_ptr())
        g.launch(1)
        torch.cuda.synchronize()
        assert torch.equal(d_out, d_ref) and float(d_ref.abs().sum()) > 0
        g.close()
        # the default stream is refused, with the context intact
        c.set_stream(0)
        with pytest.raises(ebo.EboError):
            c.record(lambda: None)
        torch.cuda.set_stream(torch.cuda.default_stream())


def test_recorded_large_edge_evaluation_takes_the_two_launch_path_and_replays(ebo, synth):
    """Round 5: a batch above eight units per CU evaluates the edge loss in two launches (compact layout, then the
    deferred units on a device-side list whose counter a hipMemsetAsync zeroes first): all three are recordable, and a
    replay -- also with other flows in the same buffer, i.e. another split into the two classes -- has the direct call's bits."""
    import torch
    ev, offsets, gt = synth.make_stream(0, 24)
    with ebo.Context(image_w=240, image_h=180, patch_w=20, patch_h=20, loss=ebo.LOSS_EDGE, tv_weight=0.0,
                     max_events=len(ev), max_windows=24) as c:
        c.set_windows(ev, offsets)
        stream = torch.cuda.Stream()
        c.set_stream(stream.cuda_stream)
        torch.cuda.set_stream(stream)
        d_flows = torch.from_numpy(gt * 0.9).to("cuda")
        d_ref = torch.zeros((24 * c.P, 3), dtype=torch.float64, device="cuda")
        d_out = torch.zeros_like(d_ref)
        c.eval_device(d_flows.data_ptr(), 1, d_ref.data_ptr())  # also sizes the tables
        torch.cuda.synchronize()
        g = c.record(lambda: c.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr()))
        g.launch(2)
        torch.cuda.synchronize()
        assert torch.equal(d_out, d_ref) and float(d_ref.abs().sum()) > 0
        d_flows.mul_(0.2)  # small boxes now: (nearly) nothing is deferred
        c.eval_device(d_flows.data_ptr(), 1, d_ref.data_ptr())
        g.launch(1)
        torch.cuda.synchronize()
        assert torch.equal(d_out, d_ref)
        g.close()
        torch.cuda.set_stream(torch.cuda.default_stream())


def test_only_device_calls_can_be_recorded(ebo, synth):
    """Anything that copies through pageable memory, allocates or synchronises would invalidate the recording
    (and on ROCm 7.2 leave the stream unusable): such entry points refuse up front, the recording goes on, and
    the context works afterwards."""
    import ctypes as C
    c, ev, offsets, gt = _ctx(ebo, synth, ebo.LOSS_VARIANCE)
    with c:
        r0, J0 = c.eval(gt * 0.5)
        lib = ebo.lib()
        h = C.c_void_p()
        assert lib.ebo_graph_end(c._h, C.byref(h)) == ebo.ERR_STATE  # end without begin
        refused = []

        def body():
            for call in (lambda: c.eval(gt * 0.5), lambda: c.set_windows(ev, offsets), lambda: c.synchronize(),
                         lambda: c.solve(ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT))):
                try:
                    call()
                except ebo.EboError as exc:
                    refused.append("recording" in str(exc))
        g = c.record(body)
        assert refused == [True] * 4
        g.launch(1)  # an empty graph
        c.synchronize()
        r1, J1 = c.eval(gt * 0.5)
        assert np.array_equal(r0, r1) and np.array_equal(J0, J1)


def test_recorded_solve_and_count_image(ebo, synth):
    """ebo_solve_device + ebo_count_image_device recorded as one step: the same flows and image as the direct calls."""
    import torch
    c, ev, offsets, gt = _ctx(ebo, synth, ebo.LOSS_VARIANCE)
    with c:
        opts = ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT)
        opts.max_num_iterations = 8
        d_sol = torch.zeros((3 * c.P, 2), dtype=torch.float64, device="cuda")
        d_stats = torch.zeros((3 * c.P, 4), dtype=torch.int32, device="cuda")
        d_img = torch.zeros((3, 180, 240), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()  # the context launches on its own stream: the fills above must have landed

        def step():
            c.solve_device(opts, d_sol.data_ptr(), d_stats.data_ptr())
            c.count_image_device(ebo.COUNT_WARPED, d_sol.data_ptr(), d_img.data_ptr())
        step()
        c.synchronize()
        ref = (d_sol.clone(), d_img.clone())
        d_sol.zero_()
        d_img.zero_()
        torch.cuda.synchronize()
        g = c.record(step)
        g.launch(2)
        c.synchronize()
        assert torch.equal(d_sol, ref[0]) and torch.equal(d_img, ref[1]) and float(d_img.sum()) > 0
