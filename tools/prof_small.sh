#!/bin/bash
# rocprofv3 kernel durations of a small launch-latency-bound configuration (config 1), to compare the
# sum of kernel times with the wall time per iteration.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_c1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c1 -- python3 $ROOT/bench.py --workload planar --particles 4 --samples 16 --traj-len 64 --dtype f64 --steps 500 --warmup 50 --no-cpu-baseline > $OUT/bench.json 2> $OUT/log.txt
cd $ROOT
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/prof_c1/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 500:
        print(r["Name"][:60], r["Calls"], r["AverageNs"]); tot += float(r["AverageNs"])
d = json.loads(open("gpurun_out/prof_c1/bench.json").read().strip().splitlines()[-1])
print("sum of kernel averages %.1f us; wall per iteration %.1f us" % (tot / 1e3, d["ms_per_step"] * 1e3))
PY
