#!/bin/bash
# Register / scratch / LDS use of the kernels in one object of ladiff_amd/csrc/build (gfx950 code object metadata).
#   scripts/kernel_regs.sh systolic [name-filter]
set -e
obj=ladiff_amd/csrc/build/$1.o
tmp=$(mktemp -d)
cp "$obj" "$tmp/o.o"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null 2>&1)
dev=$(ls "$tmp"/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$dev" | awk -v f="${2:-}" '
  /\.name:/ {name=$2}
  /\.agpr_count:/ {a=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.group_segment_fixed_size:/ {g=$2}
  /\.wavefront_size:/ { if (f == "" || index(name, f)) printf "%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s\n", name, v, a, s, p, g }'
rm -rf "$tmp"
