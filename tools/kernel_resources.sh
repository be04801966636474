#!/bin/bash
# Registers / scratch of the kernels in the built library whose name matches $1 (default: all).
# usage: bash tools/kernel_resources.sh [pattern]
set -e
T=$(mktemp -d)
cp "$(dirname "$0")/../blackbox_mpc_amd/libbbmpc.so" $T/lib.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null)
# (one offload bundle per translation unit of the library)
for B in $T/lib.so.*.hipv4-amdgcn-amd-amdhsa--gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $B; done |
  grep -E "\.name:|\.vgpr_count|\.agpr_count|\.private_segment_fixed_size|\.vgpr_spill_count|\.group_segment_fixed_size" |
  awk '/\.name:/ {name=$2} /private_segment/ {p=$2} /group_segment/ {g=$2} /\.agpr_count/ {a=$NF} /vgpr_spill/ {s=$2} /\.vgpr_count/ {print name, "vgpr", $2, "agpr", a, "scratch", p, "spills", s, "lds", g}' |
  grep -E "${1:-.}" | while read n rest; do echo "$(echo $n | c++filt | cut -c1-90)  $rest"; done
rm -rf $T
