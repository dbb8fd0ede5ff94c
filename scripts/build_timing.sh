#!/bin/bash
# build_timing.sh [name [-DFLAG ...]] -> exp/lib<name>.so (default "timing"): the regular objects with the two F(4,3) kernels
# recompiled with -DPESR_TIMING (+ ablation flags) for scripts/kernel_phases.py
set -e
cd "$(dirname "$0")/.."
mkdir -p exp
name=${1:-timing}; [ $# -gt 0 ] && shift
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-result -DPESR_TIMING $@"
hipcc $F -c pesr_amd/csrc/conv3x3_wino4.hip -o exp/$name.conv3x3_wino4.hip.o
hipcc $F -c pesr_amd/csrc/conv3x3_wgrad_wino4.hip -o exp/$name.conv3x3_wgrad_wino4.hip.o
objs=$(ls pesr_amd/build/*.o | grep -v "/conv3x3_wino4.hip.o\|/conv3x3_wgrad_wino4.hip.o")
hipcc -shared --offload-arch=gfx950 -fPIC -o exp/lib$name.so $objs exp/$name.conv3x3_wino4.hip.o exp/$name.conv3x3_wgrad_wino4.hip.o
echo exp/lib$name.so
