"""Development aid: per-kernel durations and the gaps between consecutive kernels from a rocprofv3 kernel trace (csv), for the last N launches.
    python tools/dev/trace_gaps.py <kernel_trace.csv> [N]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-N:]
dur, gap = defaultdict(list), defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = b["Kernel_Name"].split("(")[0][:60]
    dur[name].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
    gap[name].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k in dur:
    d, g = sorted(dur[k]), sorted(gap[k])
    print(f"{k:60s} n={len(d):5d} dur med {d[len(d)//2]/1e3:7.2f} us   gap-before med {g[len(g)//2]/1e3:7.2f} us")
print("span per launch (us):", (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / len(rows))
