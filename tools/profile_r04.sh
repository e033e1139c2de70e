set -u
bash tools/profile_config.sh r4_cfg3 "" > /dev/null 2>&1
bash tools/profile_config.sh r4_cfg3_sdf "--field sdf" > /dev/null 2>&1
bash tools/profile_config.sh r4_cfg3_64sph "--spheres 64" > /dev/null 2>&1
SGPMP_NO_FUSED_STEP=1 bash tools/profile_config.sh r4_cfg3_unfused "" > /dev/null 2>&1
bash tools/profile_config.sh r4_cfg2 "--workload planar" > /dev/null 2>&1
bash tools/profile_config.sh r4_cfg5 "--goals 4 --particles 512 --samples 256 --traj-len 128 --shard-of 3,8" > /dev/null 2>&1
bash tools/profile_config.sh r4_cfg1 "--workload planar --particles 4 --samples 16 --traj-len 64 --dtype f64" > /dev/null 2>&1
mkdir -p gpurun_out/r04p
SGPMP_WAVE_GROUPS=1 SGPMP_LIB_PATH=$PWD/ab_libs/stamps/libsgpmp.so python3 tools/fused_stamps.py > gpurun_out/r04p/stamps_cfg3_wave_groups.txt 2>&1
SGPMP_LIB_PATH=$PWD/ab_libs/stamps/libsgpmp.so python3 tools/fused_stamps.py > gpurun_out/r04p/stamps_cfg3_workgroups_of_256.txt 2>&1
python3 tools/dense_regime.py > gpurun_out/r04p/dense_regime.txt 2>&1
python3 tools/gpmp_bench.py > gpurun_out/r04p/gpmp_bench.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04p/gpmp_prof -o gpmp -- python3 $GRAFT_REPO_ROOT/tools/gpmp_bench.py > /dev/null 2>&1 )
cp gpurun_out/r04p/gpmp_prof/gpmp_kernel_stats.csv gpurun_out/r04p/gpmp_step_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/r04p/gpmp_prof
tools/membw > gpurun_out/r04p/membw.txt 2>&1
python3 bench.py > gpurun_out/r04p/bench_default.json 2> gpurun_out/r04p/bench_default.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04p/bench_driver_like_steps20_warmup5.json 2> /dev/null
ls gpurun_out/prof_r4_*/ | head -40; head -c 1500 gpurun_out/r04p/bench_default.json
