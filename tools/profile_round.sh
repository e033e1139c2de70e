#!/bin/bash
# Collect the profiles that bench.py's roofline block and DESIGN.md cite.  Run ON the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh v3'
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/rNN/ afterwards
# (tools/summarise_prof.py does the reduction).  PMC passes are separate runs with --kernel-trace only
# (never combined with sys/hip/hsa traces), one counter group per pass, as MI355X_MICROARCH.md says.
set -u
TAG=${1:-v3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs --no-parity"

# Per-kernel figures want whole-range launches: inside optimize(opt_iters=K) the context runs two half-range launch
# sequences that overlap each other (csrc/api.hip StepPipe), which stretches each launch's duration in a trace.
# The profiler passes therefore keep one chain; the last two bench runs are the ordinary (two-chain) ones.
export SGPMP_NO_STEP_PIPELINE=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F32"; do
    name=$(echo "$grp" | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > /dev/null 2> "$OUT/pmc_$name.log"
done
cd "$ROOT"
unset SGPMP_NO_STEP_PIPELINE
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/bench_driver_like_steps20_warmup5.json" 2> /dev/null
python3 bench.py --steps 200 --warmup 20 > "$OUT/bench_with_cpu_baseline.json" 2> "$OUT/bench.log"
python3 tools/summarise_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
