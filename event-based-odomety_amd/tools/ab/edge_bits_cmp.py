"""compares two dumps of tools/ab/edge_bits_dump.py bit for bit.  usage: edge_bits_cmp.py a.npz b.npz"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    x, y = a[k], b[k]
    same = x.view(np.uint64) == y.view(np.uint64)
    nan_both = np.isnan(x) & np.isnan(y)
    diff = ~(same | nan_both)
    if diff.any():
        bad += 1
        rel = np.abs(x - y)[diff].max() / max(np.abs(x).max(), 1e-300)
        print("%-20s %6d of %6d values differ (rows %s ...), max |d| / max |v| = %.2e" % (k, diff.sum(), diff.size, np.unique(np.nonzero(diff)[0])[:6], rel))
    else:
        print("%-20s identical (%d values, %d NaN)" % (k, x.size, np.isnan(x).sum()))
print("cases that differ:", bad)
