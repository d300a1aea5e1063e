#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` (from --kernel-trace --stats --output-format csv) into a
short CSV kept under profiles/ (kernel names truncated, all numeric columns kept)."""
import csv
import sys


def main(src, dst, note=""):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w", newline="") as f:
        if note:
            f.write(f"# {note}\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            name = r["Name"]
            name = name if len(name) <= 110 else name[:107] + "..."
            w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])


if __name__ == "__main__":
    main(*sys.argv[1:4])
