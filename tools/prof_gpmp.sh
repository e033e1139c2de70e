#!/bin/bash
# kernel-level breakdown of the Gauss-Newton planner step (tools/gpmp_bench.py under rocprofv3)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_gpmp; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o g -- python3 $ROOT/tools/gpmp_bench.py > $OUT/out.txt 2> $OUT/log.txt
cd $ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_gpmp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:70], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
cat $OUT/out.txt
