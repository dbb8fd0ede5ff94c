#!/bin/bash
# build_variant.sh <name> <file.hip> [-DFLAG ...]: recompile ONE kernel file with extra flags and link it with the
# objects of the regular build into exp/lib<name>.so (for scripts/ab_variants.py same-session A/B runs).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p exp
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result -Ipesr_amd/csrc -Iscripts/diag "$@" -c $( [ -f scripts/diag/$src ] && echo scripts/diag/$src || echo pesr_amd/csrc/$src ) -o exp/$name.$src.o
prod=$(echo $src | sed -e "s/_stag_diag\.hip$/.hip/" -e "s/_sweep_diag\.hip$/.hip/" -e "s/_vstore_diag\.hip$/.hip/" -e "s/_diag\.hip$/.hip/" -e "s/_ldsdma\.hip$/.hip/")      # a diagnostic copy replaces the product object of the same kernel file
objs=$(ls pesr_amd/build/*.o | grep -v "/$src.o" | grep -v "/$prod.o")
hipcc -shared --offload-arch=gfx950 -fPIC -o exp/lib$name.so $objs exp/$name.$src.o
echo exp/lib$name.so
