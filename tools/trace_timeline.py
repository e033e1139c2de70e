"""Timeline of the last kernels in a rocprofv3 kernel trace CSV (start / end relative to the first, per stream)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    name = r["Kernel_Name"].split("<")[0].replace("void ", "")[:22]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"q{r.get('Queue_Id', '?'):>3} {name:22s} grid {r.get('Grid_Size', '?'):>9} start {s:9.1f} us  end {e:9.1f} us  dur {e - s:7.1f}")
