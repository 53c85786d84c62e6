#!/usr/bin/env python3
"""Counters per dispatch of one kernel from a rocprofv3 --pmc run:
    python tools/pmc_kernel_table.py <dir> <kernel substring>"""
import csv
import glob
import os
import sys


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    f = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under %s" % src)
    rows = [r for r in csv.DictReader(open(f[0])) if kernel in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:90]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (disp, name), c in sorted(by.items()):
        print(disp, name, " ".join("%s=%.4g" % kv for kv in sorted(c.items())))


if __name__ == "__main__":
    main()
