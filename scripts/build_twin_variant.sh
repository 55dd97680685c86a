#!/bin/bash
# A variant of the DIAGNOSTIC TWIN for an experiment: scripts/build_twin_variant.sh <name> <file.hip> <-DDEFINE ...>
#   -> ladiff_amd/libladiff_hip_<name>.so = the twin's objects (LADIFF_STAMPS=1) with <file.hip> recompiled with the extra defines.
# Load it with LADIFF_LIB=ladiff_amd/libladiff_hip_<name>.so in the scripts that honour it.  Never shipped, never timed as the product.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -c "from ladiff_amd import build; build.build(stamps=True)" > /dev/null
stem=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DLADIFF_STAMPS=1 "$@" -c ladiff_amd/csrc/$src -o /tmp/${stem}_$name.o
objs=$(ls ladiff_amd/csrc/build_stamps/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -Wl,--version-script=ladiff_amd/csrc/exports.map -o ladiff_amd/libladiff_hip_$name.so $objs /tmp/${stem}_$name.o
echo ladiff_amd/libladiff_hip_$name.so
