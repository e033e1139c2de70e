#!/bin/bash
# Per-dispatch timeline of optimize(opt_iters=K) in its two-chain mode: rocprofv3 --kernel-trace of a short bench run, then
# tools/chain_timeline.py prints begin / end / queue of every kernel of a few iterations and the GPU-idle gaps between them.
#   gpurun --timeout 600 -- 'bash tools/chain_timeline.sh'     (outputs gpurun_out/chain_timeline.txt)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/chain_tl
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o tr -- python3 $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs --no-parity ${ARGS:-} > "$OUT/bench.json" 2> "$OUT/log.txt"
cd "$ROOT"
python3 tools/chain_timeline.py "$OUT/tr" > gpurun_out/chain_timeline.txt 2>&1
rm -rf "$OUT/tr"
tail -60 gpurun_out/chain_timeline.txt
