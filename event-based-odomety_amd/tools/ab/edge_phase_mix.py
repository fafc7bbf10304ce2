"""Per-phase dynamic instruction mix of k_eval_edge from the ablation chain's PMC summaries (edge_phase_mix.sh).
usage: edge_phase_mix.py <dir holding a0.txt a8.txt a12.txt a14.txt a15.txt>"""
import os, re, sys

def load(path, kernel):
    out, on = {}, False
    for line in open(path):
        if line.startswith("k_eval_edge"):
            on = line.startswith(kernel)
            m = re.search(r"([\d.]+) ms under", line)
            if on and m:
                out["ms"] = float(m.group(1))
        elif on and line.startswith("    "):
            k, v = line.split()
            out[k] = float(v)
    return out

d = sys.argv[1]
kernel = "k_eval_edge<true, true, 256>"
runs = {ab: load(os.path.join(d, "a%d.txt" % ab), kernel) for ab in (0, 8, 12, 14, 15)}
phases = [("gather", 0, 8), ("reverse (+zero A, convert A)", 8, 12), ("window maxima", 12, 14), ("eigenvalue runs", 14, 15),
          ("bbox+zero+scatter+convert", 15, None)]
cols = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64",
        "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS",
        "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_BRANCH", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
        "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]
print("k_eval_edge<ALIAS, JAC, 256>, reference default x 256 windows, per launch; wave-instructions in millions")
print("%-30s %8s " % ("phase", "ms") + " ".join("%9s" % c.replace("SQ_INSTS_", "").replace("SQ_", "")[:9] for c in cols))
for name, a, b in phases:
    ra, rb = runs[a], (runs[b] if b is not None else {})
    ms = ra.get("ms", 0) - rb.get("ms", 0)
    vals = [(ra.get(c, 0) - rb.get(c, 0)) / 1e6 for c in cols]
    print("%-30s %8.3f " % (name, ms) + " ".join("%9.2f" % v for v in vals))
    valu = vals[0]
    f64 = vals[1] + vals[2] + vals[3] + vals[4]
    if valu > 0:
        print("%-30s          f64 arithmetic %.0f %% of VALU (FMA %.0f, MUL %.0f, ADD %.0f, TRANS %.0f); INT32 %.0f %%, INT64 %.0f %%, CVT %.0f %%, other (moves, selects, compares, lane ops) %.0f %%"
              % ("", 100 * f64 / valu, 100 * vals[1] / valu, 100 * vals[2] / valu, 100 * vals[3] / valu, 100 * vals[4] / valu,
                 100 * vals[5] / valu, 100 * vals[6] / valu, 100 * vals[7] / valu,
                 100 * (valu - f64 - vals[5] - vals[6] - vals[7]) / valu))
ra = runs[0]
print("%-30s %8.3f " % ("whole kernel", ra.get("ms", 0)) + " ".join("%9.2f" % (ra.get(c, 0) / 1e6) for c in cols))
