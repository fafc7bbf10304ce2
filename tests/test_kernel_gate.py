"""CPU: the build-time gate on kernel shapes (event-based-odomety_amd/tools/check_kernels.py, `make
check-kernels`, run by __graft_entry__.build()).  Every kernel of the library: no dynamic stack, no
out-of-line device function or call, no generic-pointer atomics, scratch within the budget, no spilled
vector register.  And the gate is known to catch the shape that faulted on the device in round 2 (the edge
objective reached from three sites of a solver loop: outlined by the compiler, LDS arrays passed as generic
pointers, ~0.8 KB of scratch per lane): tools/probe/solve_edge_outlined.hip is compiled to assembly only --
never linked, never run -- and must be rejected for exactly those reasons."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "event-based-odomety_amd", "csrc")
TOOL = os.path.join(ROOT, "event-based-odomety_amd", "tools", "check_kernels.py")


def test_every_kernel_of_the_library_passes_the_gate():
    out = subprocess.run(["make", "-s", "-C", CSRC, "check-kernels"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 violation(s)" in out.stdout
    rep = json.loads(subprocess.run([sys.executable, TOOL, os.path.join(CSRC, "ebo_kernels.s"), "--json"],
                                    capture_output=True, text=True).stdout)
    names = list(rep["kernels"])
    assert len(names) >= 50 and not rep["outlined"]
    edge = [k for k in names if "k_eval_edge" in k or "k_solve_edge" in k]
    assert len(edge) >= 8
    for k in edge:  # the kernels the round-2 review named: spill-free now
        assert rep["kernels"][k]["vgpr_spill_stores"] == 0 and rep["kernels"][k]["calls"] == 0, k
        assert rep["kernels"][k]["scratch_bytes_per_lane"] <= 128, k


def test_the_gate_rejects_the_shape_that_faulted():
    out = subprocess.run(["make", "-s", "-C", CSRC, "check-probe"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    asm = os.path.join(ROOT, "event-based-odomety_amd", "tools", "probe", "solve_edge_outlined.s")
    rep = json.loads(subprocess.run([sys.executable, TOOL, asm, "--json"], capture_output=True, text=True).stdout)
    why = " | ".join(w for _, w in rep["violations"])
    assert any("edge_eval_outlined" in f for f in rep["outlined"])
    assert "s_swappc_b64" in why and "scratch" in why and "out-of-line device function" in why
    probe = [k for k in rep["kernels"] if "k_solve_edge_outlined" in k][0]
    assert rep["kernels"][probe]["calls"] == 3 and rep["kernels"][probe]["dynamic_stack"] == 0
