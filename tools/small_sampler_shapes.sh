#!/bin/bash
# sample_iso_small_kernel's workgroup shape (samples x waypoints per chunk) on the reference's planar example size, diagnostic
# builds: make -C stoch_gpmp_amd/csrc EXTRA=-DSGPMP_SMALL_SHAPE_ENV BUILD=$PWD/ab_libs/shape/build OUT=$PWD/ab_libs/shape
# (+ -DSGPMP_SMALL_SKIP=1: no noise phase, =2: no recurrence -> ab_libs/skip1, ab_libs/skip2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-shape}; do
export SGPMP_LIB_PATH=$PWD/ab_libs/$lib/libsgpmp.so
for sh in ${SHAPES:-8,32 8,64 4,32 4,64 2,32 2,64 1,64 16,32}; do
  SGPMP_SMALL_SHAPE=$sh timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shp -- python3 tools/example_latency.py planar 200 >/dev/null 2>&1
  f=$(find gpurun_out/shp -name "*kernel_trace.csv" | head -1)
  echo "$lib shape $sh: $(python3 tools/trace_gaps.py $f | grep sample_iso)"
  rm -rf gpurun_out/shp
done; done
