#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per (kernel, counter).

python tools/pmc_sum.py <dir> [name-filter]
"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            if flt and flt not in k:
                continue
            a = acc[(k, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        # dispatches appear once per (dimension instance); n counts rows
        print(f"{k:60s} {c:14s} rows {n:5d} sum {v:16.0f}")


if __name__ == "__main__":
    main()
