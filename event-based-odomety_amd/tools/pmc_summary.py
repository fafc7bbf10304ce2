"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs.
usage: pmc_summary.py <kernel substring> <dir> [<dir> ...]   (each dir = one -d output of one --pmc pass)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    want = sys.argv[1]
    for d in sys.argv[2:]:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        dur = defaultdict(dict)
        for f in files:
            with open(f, newline="") as fp:
                for row in csv.DictReader(fp):
                    k = row.get("Kernel_Name", "")
                    if want not in k:
                        continue
                    short = k.replace("void ", "").replace("ebo::(anonymous namespace)::", "")
                    short = short[: short.index("(")] if "(" in short else short
                    acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
                    disp[short].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
                    dur[short][row.get("Dispatch_Id", "")] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        for k, ctrs in acc.items():
            n = max(len(disp[k]), 1)
            ms = sum(dur[k].values()) / max(len(dur[k]), 1)
            print("%s  (%d dispatches, per-dispatch means, %.4f ms under the profiler) [%s]" % (k, n, ms, d))
            for c in sorted(ctrs):
                print("    %-28s %.6g" % (c, ctrs[c] / n))


if __name__ == "__main__":
    main()
