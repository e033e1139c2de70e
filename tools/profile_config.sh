#!/bin/bash
# rocprofv3 kernel stats + the PMC passes of tools/profile_round.sh for ANY bench.py configuration.  Run ON the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/profile_config.sh cfg2 "--workload planar"; bash tools/profile_config.sh cfg5 "--goals 4 --particles 512 --samples 256 --traj-len 128 --shard-of 3,8"'
#   SGPMP_NO_FUSED_STEP=1 bash tools/profile_config.sh cfg3_unfused ""       (environment switches pass through)
# Outputs under gpurun_out/prof_<tag>/ (kernel_stats.csv, pmc_summary.txt, traffic.json, bench_under_rocprof.json).
# PMC passes are separate runs with --kernel-trace only, one counter group per pass (MI355X_MICROARCH.md).
set -u
TAG=${1:?tag}
ARGS=${2:-}
STEPS=${STEPS:-60}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-other-configs --no-parity --no-sweep-alone $ARGS"
export SGPMP_NO_STEP_PIPELINE=1      # whole-range launches in every trace (see tools/profile_round.sh)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
    name=$(echo "$grp" | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > /dev/null 2> "$OUT/pmc_$name.log"
done
cd "$ROOT"
WORKLOAD="bench.py $ARGS" python3 tools/summarise_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/stats "$OUT"/pmc_*/      # (raw traces: tens of MB; the summaries above are what is kept)
tail -5 "$OUT/summary.txt"
