#!/usr/bin/env python3
"""Summaries of rocprofv3 output directories for profiles/.

  profile_summary.py stats <dir> <out.txt> [title]     top kernels of a --kernel-trace --stats run
  profile_summary.py pmc <out.json> <dir> [<dir> ...]   per-kernel mean counter values of --pmc runs (one dir per pass)

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in separate
passes, are reported in KiB, and on gfx950 FETCH_SIZE counts 128-byte requests of wide coalesced reads at 64 bytes, so
read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-byte-per-lane stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:110]


def stats(d, out, title=""):
    rows = []
    for f in find(d, "kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as fh:
        if title:
            fh.write(title + "\n")
        for r in rows[:40]:
            fh.write(f"{short(r['Name']):110s} calls={int(r['Calls']):7d} total_ms={float(r['TotalDurationNs']) / 1e6:9.1f} "
                     f"avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):5.2f}\n")
        fh.write(f"total GPU ms {total / 1e6:.3f}\n")
    print(open(out).read())


def pmc(out, dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))   # kernel -> counter -> [n, sum]
    dur = defaultdict(lambda: [0, 0.0])
    for d in dirs:
        for f in find(d, "counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    a = acc[k][r["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
                    if "Start_Timestamp" in r and r.get("End_Timestamp"):
                        t = dur[k]
                        t[0] += 1
                        t[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    res = {}
    for k, cs in acc.items():
        e = {c: {"launches": n, "mean": s / n} for c, (n, s) in cs.items()}
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            rd = 2.0 * 1024.0 * e.get("FETCH_SIZE", {"mean": 0.0})["mean"]
            wr = 1024.0 * e.get("WRITE_SIZE", {"mean": 0.0})["mean"]
            e["hbm_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr}
        if dur[k][0]:
            e["mean_duration_us_under_pmc"] = dur[k][1] / dur[k][0] / 1e3
        res[k] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k, e in sorted(res.items()):
        print(k)
        for c, v in sorted(e.items()):
            print("   ", c, v)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3:])
    else:
        raise SystemExit(__doc__)
