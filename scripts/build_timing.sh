#!/bin/bash
# build_timing.sh [name [-DFLAG ...]] -> exp/lib<name>.so (default "timing"): the regular objects with the two F(4,3) kernels
# replaced by their DIAGNOSTIC copies (scripts/diag/*_diag.hip: ablation switches, phase stamps) compiled with -DPESR_TIMING
# (+ ablation flags), for scripts/kernel_phases.py and scripts/wino4_ab.py.  The product sources carry none of this.
set -e
cd "$(dirname "$0")/.."
mkdir -p exp
name=${1:-timing}; [ $# -gt 0 ] && shift
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-result -Ipesr_amd/csrc -Iscripts/diag -DPESR_TIMING $@"
hipcc $F -c scripts/diag/conv3x3_wino4_diag.hip -o exp/$name.conv3x3_wino4.hip.o
hipcc $F -c scripts/diag/conv3x3_wgrad_wino4_diag.hip -o exp/$name.conv3x3_wgrad_wino4.hip.o
objs=$(ls pesr_amd/build/*.o | grep -v "/conv3x3_wino4.hip.o\|/conv3x3_wgrad_wino4.hip.o")
hipcc -shared --offload-arch=gfx950 -fPIC -o exp/lib$name.so $objs exp/$name.conv3x3_wino4.hip.o exp/$name.conv3x3_wgrad_wino4.hip.o
echo exp/lib$name.so
