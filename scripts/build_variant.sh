#!/bin/bash
# bash scripts/build_variant.sh NAME [-DTOHC_...=...]  -> scripts/native/libtohc_NAME.so, an experiment build of the library
# (engine.hip with extra flags; sort.o / shape_hull.o from the product build).  Used with TOHC_LIB=... / scripts/bench_variants.sh.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p scripts/native /tmp/tohc_var
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 "$@" -c trajopt_amd/csrc/engine.hip -o /tmp/tohc_var/engine_$NAME.o \
  -Rpass-analysis=kernel-resource-usage 2> /tmp/tohc_var/res_$NAME.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared /tmp/tohc_var/engine_$NAME.o trajopt_amd/csrc/_obj/sort.o trajopt_amd/csrc/_obj/shape_hull.o -o scripts/native/libtohc_$NAME.so
echo "built scripts/native/libtohc_$NAME.so"
