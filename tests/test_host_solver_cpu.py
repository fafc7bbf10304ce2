"""CPU test of the PRODUCT's host trust-region solver (csrc/host_lm.cpp) without a GPU: the
data-term residuals/Jacobians it asks for are supplied by the oracle instead of the device, and
the outcome is compared with the oracle's own solver on the reference's problem (data + TV terms,
feature_detector.cpp:316-414).  Same evaluations in, so the flows agree to rounding of the
linear algebra (banded vs dense Cholesky)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


@pytest.fixture(scope="module")
def shim(ebo):
    subprocess.check_call(["make", "-s", "-C", CPP, "libhostlm_shim.so"])
    lib = C.CDLL(os.path.join(CPP, "libhostlm_shim.so"))
    lib.hlm_create.restype = C.c_void_p
    lib.hlm_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    for f in (lib.hlm_request, lib.hlm_supply, lib.hlm_result, lib.hlm_stats, lib.hlm_destroy):
        f.argtypes = None
    lib.hlm_request.argtypes = [C.c_void_p, C.c_void_p]
    lib.hlm_supply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hlm_result.argtypes = [C.c_void_p, C.c_void_p]
    lib.hlm_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hlm_destroy.argtypes = [C.c_void_p]
    return lib


def run_host_lm(shim, ebo, orc, ev, prm, opts):
    npx, npy = orc.grid(prm)
    P = npx * npy
    _, _, active, _ = orc.window_eval(ev, prm, np.zeros((P, 2)), want_jac=False)
    act = active.astype(np.uint8)
    h = shim.hlm_create(npx, npy, act.ctypes.data_as(C.c_void_p), prm.tv_weight, prm.tv_huber, C.byref(opts))
    flows = np.zeros((P, 2))
    rounds = 0
    while True:
        q = shim.hlm_request(h, flows.ctypes.data_as(C.c_void_p))
        if q == 0:
            break
        r, J, _, _ = orc.window_eval(ev, prm, flows, want_jac=(q == 1))
        Jp = J.ctypes.data_as(C.c_void_p) if q == 1 else None
        shim.hlm_supply(h, r.ctypes.data_as(C.c_void_p), Jp)
        rounds += 1
    out = np.zeros((P, 2))
    shim.hlm_result(h, out.ctypes.data_as(C.c_void_p))
    st = np.zeros(4, dtype=np.int32)
    costs = np.zeros(2)
    shim.hlm_stats(h, st.ctypes.data_as(C.c_void_p), costs.ctypes.data_as(C.c_void_p))
    shim.hlm_destroy(h)
    return out, st, costs, rounds


@pytest.mark.parametrize("loss,tv", [(1, 1e3), (1, 0.0), (0, 1e3)])
def test_product_host_lm_matches_oracle_solver(shim, ebo, orc, synth, loss, tv):
    n = 6000 if loss == 1 else 2500
    ev, _ = synth.make_window(0, n_events=n)
    iters = 50 if tv else 14
    prm = orc.default_params(loss=loss, tv_weight=tv)
    opts = ebo.default_solver(max_num_iterations=iters)
    flows, st, costs, rounds = run_host_lm(shim, ebo, orc, ev, prm, opts)
    fo, _, so = orc.compensate_events_contrast(ev, prm, orc.default_solver(max_num_iterations=iters), want_image=False)
    assert np.abs(flows - fo).max() <= 1e-9
    assert st[0] == so.iterations and st[3] == so.termination
    assert costs[0] == pytest.approx(so.initial_cost, rel=1e-14)
    assert costs[1] == pytest.approx(so.final_cost, rel=1e-10)
    assert rounds == st[1] + st[2]  # one batched evaluation per request


def test_product_host_lm_degenerate_problems(shim, ebo, orc, synth):
    """No active patch and no TV: nothing to solve; with TV only: stays at zero."""
    ev, _ = synth.make_window(0, n_events=300)  # < 100 events in every patch
    for tv in (0.0, 1e3):
        prm = orc.default_params(loss=1, tv_weight=tv)
        flows, st, costs, rounds = run_host_lm(shim, ebo, orc, ev, prm, ebo.default_solver())
        assert np.all(flows == 0.0)
        assert st[3] == 0
        fo, _, so = orc.compensate_events_contrast(ev, prm, orc.default_solver(), want_image=False)
        assert np.all(fo == 0.0)


def test_converged_solves_wander_at_noise_level(orc, synth):
    """CPU only, the oracle against ITSELF: the reference-default call (edge loss, TV, 50 iterations) on
    two windows whose solve ends in a flat valley.  Moving compensateScale by 1e-12 relative -- the size
    of the sum-order differences between any two correct implementations of the objective (the HIP
    path's r differs from the oracle's by ~2e-12) -- leaves the final cost where it was (<= 1e-10
    relative) but changes the iteration count and / or moves the flows by more than 1e-5: such windows have
    no answer to 1e-5, which is why tests/test_gpu_edge_ties.py asserts the trajectory on ~90 % of the
    windows and the converged cost on all of them."""
    moved = 0
    for window, factors in ((130, (1 + 1e-12,)), (190, (1 - 1e-12, 1 + 3e-12))):
        ev, _ = synth.make_window(0, window=window, n_events=15000)
        base, _, sb = orc.compensate_events_contrast(ev, orc.default_params(loss=0), orc.default_solver(), want_image=False)
        for f in factors:
            got, _, sg = orc.compensate_events_contrast(ev, orc.default_params(loss=0, scale=1e-3 * f), orc.default_solver(),
                                                        want_image=False)
            assert abs(sg.final_cost - sb.final_cost) <= 1e-10 * sb.final_cost
            d = float(np.abs(got - base).max())
            assert d <= 1e-3
            if d > 1e-5 or sg.iterations != sb.iterations:
                moved += 1
    assert moved >= 3


def test_abi_host_solver_state_machine(ebo, orc, synth):
    """ebo_lm_* of the C ABI (the same HostLm, exported for callers that put a collective between "evaluate" and
    "step": SURVEY 8(e)'s reference-faithful TV mode across GPUs), driven on the CPU with the oracle's data terms:
    the oracle solver's flows, iteration count and termination; argument and state errors."""
    ev, _ = synth.make_window(0, n_events=2500)
    prm = orc.default_params(loss=0, tv_weight=1e3)
    npx, npy = orc.grid(prm)
    P = npx * npy
    _, _, active, _ = orc.window_eval(ev, prm, np.zeros((P, 2)), want_jac=False)
    opts = ebo.default_solver(max_num_iterations=50)
    with ebo.HostSolver(npx, npy, active, tv_weight=prm.tv_weight, tv_huber=prm.tv_huber, opts=opts) as lm:
        rounds = 0
        while True:
            what, flows = lm.request()
            if what == ebo.HostSolver.DONE:
                break
            again, flows2 = lm.request()  # asking twice changes nothing
            assert again == what and np.array_equal(flows, flows2)
            r, J, _, _ = orc.window_eval(ev, prm, flows, want_jac=(what == ebo.HostSolver.NEED_JACOBIAN))
            if what == ebo.HostSolver.NEED_JACOBIAN:
                with pytest.raises(ebo.EboError) as err:  # a Jacobian was asked for
                    lm.supply(r, None)
                assert err.value.code == ebo.ERR_ARG
            lm.supply(r, J)
            rounds += 1
        out, summ = lm.result()
        with pytest.raises(ebo.EboError) as err:  # the solve has finished
            lm.supply(np.zeros(P), np.zeros((P, 2)))
        assert err.value.code == ebo.ERR_STATE
    fo, _, so = orc.compensate_events_contrast(ev, prm, orc.default_solver(max_num_iterations=50), want_image=False)
    assert np.abs(out - fo).max() <= 1e-9
    assert summ.iterations == so.iterations and summ.termination == so.termination
    assert rounds == summ.num_evals_cost + summ.num_evals_jac
    assert summ.final_cost == pytest.approx(so.final_cost, rel=1e-9)
    with pytest.raises(ebo.EboError):
        ebo.HostSolver(0, 3, np.zeros(0))


def test_variance_loss_with_tv_starts_in_a_spurious_minimum_on_the_cpu_alone(orc, synth):
    """Why the north-star objective under the reference's own coupling (variance loss + TV 1e3, start at zero flow,
    feature_detector.cpp:318-326) returns its start point -- shown on the CPU oracle alone, no device involved.
    At exactly zero flow every event sits on an integer position, and compensateEvents truncates the warped
    coordinate (`int(c)`, contrast_functor.h:59,63): an ARBITRARILY small flow puts the events it moves in the
    negative direction into the bin below, their 7x7 splat window shifts by a pixel, the image changes by a finite
    amount and the variance drops.  So the cost JUMPS up by the same finite amount for a step of 1e-12 and of 1e-9 --
    a discontinuity, not a slope -- although the gradient direction is a descent direction further out.  Every LM
    step lands on the upper side of the jump and is rejected until the trust region is ~1e-17: the solve ends where
    it started.  (The edge loss has the same truncation but a gain that outruns the jump within the trust region.)"""
    ev, _ = synth.make_window(0, n_events=15000)
    prm = orc.default_params(loss=1)  # variance loss, compensateTVweight 1e3
    flows, _, summ = orc.compensate_events_contrast(ev, prm, orc.default_solver(), want_image=False)
    assert summ.iterations >= 10 and np.abs(flows).max() < 1e-12
    P = flows.shape[0]
    r0, J0, active, _ = orc.window_eval(ev, prm, np.zeros((P, 2)))
    g = r0[:, None] * J0
    d = -g / np.abs(g).max()  # steepest descent of the data terms, largest component 1

    def dcost(step):
        r, _, _, _ = orc.window_eval(ev, prm, step * d)
        return 0.5 * float((r ** 2).sum() - (r0 ** 2).sum())
    tiny, small, far = dcost(1e-12), dcost(1e-9), dcost(1e-2)
    print("data cost against zero flow: step 1e-12 %+.3f, 1e-9 %+.3f, 1e-2 %+.3f" % (tiny, small, far))
    assert tiny > 1.0 and abs(small - tiny) < 0.1 * tiny  # the same jump whatever the step: a discontinuity
    assert far < 0.0                                      # and a real descent direction beyond it
    # the slope the Jacobian promises is orders of magnitude below the jump for any step LM would accept
    predicted = float((g * (1e-9 * d)).sum())
    assert abs(predicted) < 1e-3 * tiny
