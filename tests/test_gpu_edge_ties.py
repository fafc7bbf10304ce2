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
per-window table (tools/edge_tie_table.py).

What IS asserted on every window, wandering or not (round 3) -- the solvers' own equivalence:
  * the ORACLE's objective (data terms + Huber-wrapped TV terms, feature_detector.cpp:357-396) at the HIP
    flows equals the oracle's objective at its own flows to 1e-10 relative, and the HIP path's objective at
    the oracle's flows equals the HIP objective at its own flows to 1e-10 (observed <= 1.2e-11 on the
    wandering windows, 0 .. 4e-16 on the others; function_tolerance is 1e-12 per step): neither solver can
    tell the two end points apart by the quantity it minimises;
  * the gradient at the other side's end point is at the level the solver stopped at itself (max-norm
    within 10x of its own, which is 0.4 .. 270 against costs of ~5e7: both stop on function_tolerance);
  * 92 of the 100 walk the oracle's trajectory within 1e-5 (measured 93)."""
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
    assert len(same) >= 92  # measured 93 (round 2 and round 3)
    # the test's own restatement of the objective is the oracle solver's: its cost at its own end point
    assert max(r["cost_formula_rel"] for r in rows) <= 1e-13
    # the solvers' own equivalence on EVERY window (the wandering ones are where it says something)
    wander = [r for r in rows if r not in same]
    print("wandering windows: " + ", ".join("%d (%d/%d its, flows %.1e apart, cross-cost %.1e / %.1e)"
                                            % (r["window"], r["iterations"], r["iterations_oracle"], r["max_dflow"],
                                               r["cross_cost_rel_oracle"], r["cross_cost_rel_hip"]) for r in wander))
    for r in rows:
        assert r["cross_cost_rel_oracle"] <= 1e-10 and r["cross_cost_rel_hip"] <= 1e-10, r
        assert r["grad_oracle_at_hip"] <= 10.0 * max(r["grad_oracle_at_oracle"], 1.0), r
        assert r["grad_hip_at_oracle"] <= 10.0 * max(r["grad_hip_at_hip"], 1.0), r
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
