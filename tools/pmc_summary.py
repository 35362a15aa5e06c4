#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, per counter, value summed over the rows of
one dispatch (one row per XCD / dimension instance) and averaged over dispatches."""
import collections
import csv
import glob
import sys

def summarise(root, match=()):
    res = collections.defaultdict(dict)
    for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            per[(r['Kernel_Name'][:70] + ' grid=' + r['Grid_Size'], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
        for (k, c), d in per.items():
            if match and not any(m in k for m in match):
                continue
            v = list(d.values())
            res[k][c] = sum(v) / len(v)
    return res

if __name__ == '__main__':
    res = summarise(sys.argv[1], sys.argv[2:])
    for k, cs in res.items():
        print(k[:100])
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} {v:16.0f}")
