#!/bin/bash
# rocprofv3 stats + a few PMC groups for the planar workload (config 2).  Run ON the GPU box.
set -u
TAG=${1:-planar}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload planar --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo "$grp" | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > /dev/null 2> "$OUT/pmc_$name.log"
done
cd "$ROOT"
python3 tools/summarise_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
