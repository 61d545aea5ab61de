"""Per-kernel means of rocprofv3 counter-collection CSVs (one row per dispatch and counter).

    python tools/summarize_pmc.py gpurun_out/pmc_sq [gpurun_out/pmc_fetch ...] > profiles/rNN_pmc_summary.txt
"""
import collections
import csv
import glob
import os
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -sum(len(v) for v in acc[k].values())):
        print(k[:150])
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-28s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
