# Round 6: rocprofv3 kernel stats + PMC passes of every single-GPU configuration and mode (run ON the GPU box):
#   gpurun --timeout 3000 -- 'bash tools/profile_r06.sh'
# then   python tools/collect_traffic.py r06 final cfg3=r6_cfg3 cfg3_store_free=r6_cfg3_store_free cfg3_unfused=r6_cfg3_unfused ...
set -u
NS="--no-store-free"
C5="--goals 4 --particles 512 --samples 256 --traj-len 128 --shard-of 3,8"
bash tools/profile_config.sh r6_cfg3 "$NS" > /dev/null 2>&1
bash tools/profile_config.sh r6_cfg3_store_free "--store-free" > /dev/null 2>&1
# the stand-alone sampler and sweep (north_star's 40 % clause is about the sweep as a launch of its own)
SGPMP_NO_FUSED_STEP=1 bash tools/profile_config.sh r6_cfg3_unfused "$NS" > /dev/null 2>&1
bash tools/profile_config.sh r6_cfg2 "--workload planar $NS" > /dev/null 2>&1
bash tools/profile_config.sh r6_cfg2_store_free "--workload planar --store-free" > /dev/null 2>&1
bash tools/profile_config.sh r6_cfg5 "$C5 $NS" > /dev/null 2>&1
bash tools/profile_config.sh r6_cfg5_store_free "$C5 --store-free" > /dev/null 2>&1
bash tools/profile_config.sh r6_cfg1 "--workload planar --particles 4 --samples 16 --traj-len 64 --dtype f64 $NS" > /dev/null 2>&1
# fp64 at config 3's shape: the one-launch step (round 6), its fp32-link-fields option, and the two-launch step it replaces
STEPS=20 bash tools/profile_config.sh r6_cfg3_f64 "--dtype f64 $NS" > /dev/null 2>&1
STEPS=20 SGPMP_F64_FIELDS_F32=1 bash tools/profile_config.sh r6_cfg3_f64_mixed "--dtype f64 $NS" > /dev/null 2>&1
STEPS=20 SGPMP_NO_FUSED_STEP=1 bash tools/profile_config.sh r6_cfg3_f64_unfused "--dtype f64 $NS" > /dev/null 2>&1
mkdir -p gpurun_out/r06
tools/membw > gpurun_out/r06/membw.txt 2>&1
for t in r6_cfg3 r6_cfg3_store_free r6_cfg3_unfused r6_cfg2 r6_cfg2_store_free r6_cfg5 r6_cfg5_store_free r6_cfg1 r6_cfg3_f64 r6_cfg3_f64_mixed r6_cfg3_f64_unfused; do
  echo "== $t"; head -4 gpurun_out/prof_$t/kernel_stats.csv | cut -c1-160
done
