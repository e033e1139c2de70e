#!/bin/bash
# tools/pipeline_ab.py (two chains vs single-iteration calls) for several builds of the library on one box
for lib in "$@"; do echo "== $lib"; SGPMP_LIB_PATH=$PWD/$lib python3 tools/pipeline_ab.py 300 2>&1 | grep -v amdgpu.ids | head -2; done
