# Round 5: rocprofv3 kernel stats + PMC passes of every single-GPU configuration, storing AND store-free (run ON the GPU box):
#   gpurun --timeout 3000 -- 'bash tools/profile_r05.sh'     then   python tools/collect_traffic.py r05 final cfg3=r5_cfg3 ...
set -u
NS="--no-store-free"
bash tools/profile_config.sh r5_cfg3 "$NS" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg3_store_free "--store-free" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg3_sdf "--field sdf $NS" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg3_sdf_store_free "--field sdf --store-free" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg3_64sph "--spheres 64 $NS" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg3_64sph_store_free "--spheres 64 --store-free" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg2 "--workload planar $NS" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg5 "--goals 4 --particles 512 --samples 256 --traj-len 128 --shard-of 3,8 $NS" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg5_store_free "--goals 4 --particles 512 --samples 256 --traj-len 128 --shard-of 3,8 --store-free" > /dev/null 2>&1
bash tools/profile_config.sh r5_cfg1 "--workload planar --particles 4 --samples 16 --traj-len 64 --dtype f64 $NS" > /dev/null 2>&1
mkdir -p gpurun_out/r05p
python3 tools/store_free_ab.py > gpurun_out/r05p/store_free_ab.txt 2>&1
tools/membw > gpurun_out/r05p/membw.txt 2>&1
ls gpurun_out/prof_r5_*/ | head -60
