"""GPU test of the framework-free RCCL exchange (ebo_comm_*, ebo_allgather_device) at the one
rank a single-GPU box allows: id creation, communicator init, an all-gather of the device
results of a real evaluation on the context's stream, teardown.  The N > 1 layout of the gather
is rehearsed on CPU by tests/test_multiprocess.py."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allgather_single_rank(ebo, synth):
    ev, gt = synth.make_window(0, n_events=15000)
    hip = ctypes.CDLL("libamdhip64.so")
    with ebo.Context(loss=ebo.LOSS_VARIANCE) as c:
        c.set_window(ev)
        n = c.P
        d_flows, d_out, d_all = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d_flows), ctypes.c_size_t(n * 16)) == 0
        assert hip.hipMalloc(ctypes.byref(d_out), ctypes.c_size_t(n * 24)) == 0
        assert hip.hipMalloc(ctypes.byref(d_all), ctypes.c_size_t(n * 24)) == 0
        flows = np.ascontiguousarray(gt * 0.5)
        assert hip.hipMemcpy(d_flows, flows.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * 16), 1) == 0
        with pytest.raises(ebo.EboError) as ei:
            c.allgather_device(d_out.value, d_all.value, n * 3)
        assert ei.value.code == ebo.ERR_STATE  # no communicator yet
        cid = ebo.comm_unique_id()
        assert len(cid) == 128 and any(cid)
        c.comm_init(cid, 0, 1)
        c.eval_device(d_flows.value, 1, d_out.value)      # async on the context's stream
        c.allgather_device(d_out.value, d_all.value, n * 3)  # ordered after it on the same stream
        c.synchronize()
        got = np.zeros((n, 3))
        assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_all, ctypes.c_size_t(n * 24), 2) == 0
        r, J = c.eval(flows)
        assert np.array_equal(got[:, 0], r[0]) and np.array_equal(got[:, 1:], J[0])
        c.comm_destroy()
        c.comm_destroy()  # idempotent
        for p in (d_flows, d_out, d_all):
            hip.hipFree(p)
