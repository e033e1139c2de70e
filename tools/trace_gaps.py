"""Per-kernel durations and the gaps between consecutive kernels from a rocprofv3 kernel trace CSV (last 60 % of the trace)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.4):]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0][:60]
    dur[k] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    gap[k] += int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    cnt[k] += 1
for k in sorted(cnt, key=lambda k: -dur[k]):
    print(f"{k:62s} n={cnt[k]:5d} dur={dur[k] / cnt[k] / 1e3:8.2f} us  gap_after={gap[k] / cnt[k] / 1e3:8.2f} us")
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span per kernel:", span / len(rows) / 1e3, "us over", len(rows))
