#!/bin/bash
# A one-off variant of the library for a timing probe: scripts/build_variant.sh <name> <-DDEFINE ...>  ->  ladiff_amd/libladiff_hip_<name>.so
# (systolic.hip compiled with the extra defines, every other object taken from the product build).  Results of such a build are garbage by
# design; load it with LADIFF_LIB=ladiff_amd/libladiff_hip_<name>.so scripts/handoff_ab.py ...   Never shipped, never timed as the product.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
python -m ladiff_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -c ladiff_amd/csrc/systolic.hip -o /tmp/systolic_$name.o
objs=$(ls ladiff_amd/csrc/build/*.o | grep -v systolic.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -Wl,--version-script=ladiff_amd/csrc/exports.map -o ladiff_amd/libladiff_hip_$name.so $objs /tmp/systolic_$name.o
echo ladiff_amd/libladiff_hip_$name.so
