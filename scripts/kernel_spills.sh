#!/bin/bash
# Where a kernel of one object spills: scripts/kernel_spills.sh <object stem> <kernel-name substring>  -> line numbers of scratch ops in
# the kernel's disassembly (written to /tmp/dis_<stem>.s), with counts of MFMA / LDS-counter instructions per 500-line window for orientation
set -e
obj=ladiff_amd/csrc/build/$1.o
tmp=$(mktemp -d)
cp "$obj" "$tmp/o.o"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null 2>&1)
dev=$(ls "$tmp"/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn "$dev" | awk -v k="$2" '/^[0-9a-f]+ </{f = index($0, k) > 0} f{print}' > /tmp/dis_$1.s
rm -rf "$tmp"
wc -l /tmp/dis_$1.s
echo "scratch ops per 500-line window:"; grep -n "scratch_" /tmp/dis_$1.s | awk -F: '{print int($1/500)*500}' | uniq -c | tr '\n' ' '; echo
echo "ds_add_u32 (v2 roles) at:"; grep -n "ds_add_u32" /tmp/dis_$1.s | awk -F: '{print $1}' | tr '\n' ' '; echo
echo "v_writelane: $(grep -c v_writelane /tmp/dis_$1.s)  v_readlane: $(grep -c v_readlane /tmp/dis_$1.s)"
