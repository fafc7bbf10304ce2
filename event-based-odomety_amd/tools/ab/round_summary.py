"""Sums the per-round trace of a pipelined lock-step solve (EBO_SOLVE_TRACE=1 EBO_SOLVE_TRACE_ROUNDS=1, stderr) by phase of
the solve: rounds, window-evaluations, time waiting for the device, in the LM steps, in the requests.
usage: round_summary.py <log>   (the last solve of the log is summarised)"""
import re
import sys

rows, solves = [], []
for line in open(sys.argv[1]):
    m = re.search(r"round\s+(\d+) group (\d) live\s+(\d+) jac (\d): wait ([\d.]+) supply ([\d.]+) request ([\d.]+)", line)
    if m:
        rows.append(tuple(float(x) for x in m.groups()))
    if "lock-step solve" in line:
        solves.append(rows)
        rows = []
rows = solves[-1]
for a, b in ((0, 20), (20, 60), (60, 110), (110, 10000)):
    r = [x for x in rows if a <= x[0] < b]
    print("rounds %3d-%-5d: n=%3d window-evaluations %5d  wait %.2f  supply %.2f  request %.2f ms"
          % (a, b, len(r), sum(x[2] for x in r), sum(x[4] for x in r), sum(x[5] for x in r), sum(x[6] for x in r)))
print("all: wait %.2f supply %.2f request %.2f ms" % (sum(x[4] for x in rows), sum(x[5] for x in rows), sum(x[6] for x in rows)))
