"""The reference-default call end to end on 100 seeded windows (edge loss, TV-coupled global LM,
50 iterations; every solve starts at exactly zero flow, where the edge loss can have exact ties --
windows whose first Jacobian differs from the oracle's on some patch are counted).

What is asserted, and why not "1e-5 on all 100": the objective value and Jacobian agree with the
oracle to ~2e-12 / 1e-11 at every iterate (sum order), and on ~93 % of the windows the two solvers
walk the same trajectory (same iteration count, flows within 1e-5, observed <= 5e-6).  On the rest
both have CONVERGED -- final costs equal to <= 1e-10 relative -- but in the flat end-game a step
whose predicted decrease is at the noise level of the cost (|r| ~ 1e3, differences of 1e-12) is
accepted by one and rejected by the other, the iteration counts drift apart and the flows end up
to 1.4e-4 apart along the valley floor.  The CPU oracle does the same to ITSELF when its inputs
move by 1e-12 relative (tests/test_host_solver_cpu.py::test_converged_solves_wander_at_noise_level,
CPU only), so those windows have no answer to 1e-5.  profiles/r02_edge_tie_table.md is the
per-window table (tools/edge_tie_table.py)."""
import pytest

import edge_ties

pytestmark = pytest.mark.gpu


def test_reference_default_call_on_100_windows(ebo, orc, synth):
    rows = edge_ties.run(ebo, orc, synth, 100)
    assert len(rows) == 100
    ties = sum(r["tie_patches"] for r in rows)
    same = [r for r in rows if r["iterations"] == r["iterations_oracle"] and r["termination"] == r["termination_oracle"]
            and r["max_dflow"] <= 1e-5]
    print("same trajectory and flows within 1e-5: %d of 100; zero-flow tie patches: %d of %d active; largest flow "
          "difference %.2e; largest relative final-cost difference %.1e"
          % (len(same), ties, sum(r["active"] for r in rows), max(r["max_dflow"] for r in rows),
             max(r["final_cost_rel"] for r in rows)))
    assert all(r["value_ok"] for r in rows)
    assert len(same) >= 88
    # every window, also the ones whose end-game differs: the same minimum
    assert max(r["final_cost_rel"] for r in rows) <= 1e-9
    assert max(r["max_dflow"] for r in rows) <= 1e-3
    # the termination kind belongs to the trajectory (a window that wanders can run into the
    # iteration cap in one solver and meet a tolerance just before it in the other): it is part of
    # `same` above; on the wandering windows only the minimum is asserted
    odd = [r for r in rows if r["termination"] != r["termination_oracle"]]
    for r in odd:
        print("termination differs: window %d, %d / %d iterations, termination %d / %d, flows %.2e apart, cost %.1e"
              % (r["window"], r["iterations"], r["iterations_oracle"], r["termination"], r["termination_oracle"],
                 r["max_dflow"], r["final_cost_rel"]))
    assert len(odd) <= 3
