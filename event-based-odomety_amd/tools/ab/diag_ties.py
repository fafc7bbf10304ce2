import sys, importlib, numpy as np
import os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
import orc, edge_ties
ebo = importlib.import_module("event-based-odomety_amd"); synth = importlib.import_module("event-based-odomety_amd.synth")
rows = edge_ties.run(ebo, orc, synth, 100)
for r in rows:
    wander = not (r["iterations"] == r["iterations_oracle"] and r["max_dflow"] <= 1e-5)
    if wander or r["window"] % 20 == 0:
        print(r["window"], "W" if wander else " ", r["iterations"], r["iterations_oracle"], "dflow %.1e" % r["max_dflow"], "formula %.1e" % r["cost_formula_rel"],
              "xcost orc %.1e hip %.1e" % (r["cross_cost_rel_oracle"], r["cross_cost_rel_hip"]),
              "grad o@h %.2e o@o %.2e h@h %.2e h@o %.2e" % (r["grad_oracle_at_hip"], r["grad_oracle_at_oracle"], r["grad_hip_at_hip"], r["grad_hip_at_oracle"]))
print("max formula", max(r["cost_formula_rel"] for r in rows), "max xcost", max(max(r["cross_cost_rel_oracle"], r["cross_cost_rel_hip"]) for r in rows))
print("max grads", max(r["grad_oracle_at_hip"] for r in rows), max(r["grad_oracle_at_oracle"] for r in rows))
