#!/usr/bin/env python3
"""Print mean counter values per kernel from rocprofv3 --pmc output directories (development aid).

  python tools/pmc_table.py gpurun_out/pmc_a gpurun_out/pmc_b ... [--match substring]
"""
import collections
import csv
import glob
import os
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    if match in args:
        args.remove(match)
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    for src in args:
        for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0]
                if match in k:
                    table[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in sorted(table.items()):
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:36s} mean {sum(v)/len(v):16.1f}  n={len(v)}")


if __name__ == "__main__":
    main()
