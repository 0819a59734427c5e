#!/usr/bin/env python3
"""Per-kernel means of arbitrary rocprofv3 --pmc counters: pmc_sq.py out.json pass1.csv [pass2.csv ...]"""
import collections
import csv
import json
import sys


def short(name):
    return name.split("(")[0].replace("void sefd::", "").replace("sefd::", "")


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    res = {k: {c: v / max(1, len(n[k][c])) for c, v in cs.items()} for k, cs in acc.items() if not (k.startswith("at::") or k.startswith("__amd"))}
    for k, cs in res.items():
        cs["_launches"] = max(len(s) for s in n[k].values())
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, cs in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0)) * kv[1]["_launches"]):
        print(k[:70], {c: round(v) for c, v in cs.items()})


if __name__ == "__main__":
    main()
