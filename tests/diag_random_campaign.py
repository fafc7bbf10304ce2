"""One-off wider sweep of tests/test_gpu_random.py's cases on the GPU box (seeds beyond the 24 the
suite runs): python tests/diag_random_campaign.py <first> <last>.  Prints the failing seeds."""
import importlib
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
ebo = importlib.import_module("event-based-odomety_amd")
import orc  # noqa: E402
import test_gpu_random as T  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    try:
        T.test_random_windows_match_the_oracle(ebo, orc, seed)
    except Exception:  # report and go on
        bad.append(seed)
        print("seed %d FAILED" % seed)
        traceback.print_exc(limit=3)
    if seed % 20 == 0:
        print("... seed %d done, %d failures so far" % (seed, len(bad)), flush=True)
print("seeds [%d, %d): %d failures %s" % (first, last, len(bad), bad))
