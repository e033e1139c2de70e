"""Timeline of the kernels of an `optimize(opt_iters=K)` call from a rocprofv3 --kernel-trace csv (tools/chain_timeline.sh)."""
import csv, glob, sys, collections

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""),
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
fused = [r for r in rows if "fused_step" in r[2]]
print(len(rows), "dispatches,", len(fused), "fused launches; queues:", collections.Counter(r[3] for r in fused))
# the longest run of half-launches = the timed optimize(opt_iters=K) call: take the last 24 fused launches but 4
# the two-chain launches run on their own two queues (the ones with the most fused launches); single-iteration calls on another
qs = [q for q, _ in collections.Counter(r[3] for r in fused).most_common(2)]
two = [r for r in fused if r[3] in qs]
win = two[len(two) // 2: len(two) // 2 + 16]
t0 = win[0][0]
sel = [r for r in rows if win[0][0] <= r[0] <= win[-1][1]]
print("window of", len(win), "fused launches; times in us from the window start")
busy_end = None
idle = 0.0
for s, e, n, q, st in sel:
    gap = "" if busy_end is None or s <= busy_end else f"   <-- GPU idle {(s - busy_end) / 1e3:6.1f} us"
    if busy_end is not None and s > busy_end:
        idle += (s - busy_end) / 1e3
    busy_end = e if busy_end is None else max(busy_end, e)
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:6.1f} us)  q{q:>3} {n}{gap}")
span = (sel[-1][1] - t0) / 1e3
print(f"span {span:.1f} us, GPU idle (no kernel resident) {idle:.1f} us")
# per queue: gap between a kernel's end and the next kernel's start on the same queue
byq = collections.defaultdict(list)
for r in sel:
    byq[r[3]].append(r)
for q, rs in byq.items():
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(rs, rs[1:])]
    if gaps:
        print(f"queue {q}: {len(rs)} kernels, in-queue gaps mean {sum(gaps) / len(gaps):.1f} us, max {max(gaps):.1f}")
dur = [(r[1] - r[0]) / 1e3 for r in win]
print(f"fused half-launch duration mean {sum(dur) / len(dur):.1f} us; iteration period {(win[-1][0] - win[0][0]) / 1e3 / (len(win) - 1) * 2:.1f} us")
