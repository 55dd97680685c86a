#!/bin/bash
# A variant of the PRODUCT library in which ONE source file is replaced: scripts/build_file_variant.sh <name> <stem> <path/to/replacement.hip> [-DDEFINE ...]
#   -> ladiff_amd/libladiff_hip_<name>.so (every other object from the product build).  For same-box A/B runs of an older or experimental
# form of one kernel file (e.g. `git show HEAD~3:ladiff_amd/csrc/dec_mlp.hip > /tmp/dec_mlp_old.hip`); load with LADIFF_LIB=... in the
# scripts that honour it.  Never shipped.
set -e
name=$1; stem=$2; src=$3; shift 3
cd "$(dirname "$0")/.."
python -m ladiff_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I ladiff_amd/csrc -I include "$@" -c $src -o /tmp/${stem}_$name.o
objs=$(ls ladiff_amd/csrc/build/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -Wl,--version-script=ladiff_amd/csrc/exports.map -o ladiff_amd/libladiff_hip_$name.so $objs /tmp/${stem}_$name.o
echo ladiff_amd/libladiff_hip_$name.so
