#!/bin/bash
# Register / spill / scratch / LDS figures of every kernel in an object file or shared library built by hipcc (what
# llvm-readelf --notes says about the gfx950 code object inside).  usage: tools/codeobj.sh hast_amd/csrc/gz_kernels.o [name filter]
set -e
in=$(readlink -f "$1"); filt=${2:-.}
tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
cp "$in" "$tmp/x.o"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1)
co=$(ls "$tmp"/x.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | awk -v f="$filt" '
  /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2}
  /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2}
  /\.vgpr_spill_count:/ {vs=$2; if (name ~ f) printf "%-90s vgpr %3d sgpr %3d vgpr_spill %3d sgpr_spill %3d scratch %5d lds %6d\n", name, vg, sg, vs, ss, scr, lds}'
# with DISASM=out.s in the environment the code object's disassembly is written there as well
if [ -n "$DISASM" ]; then /opt/rocm/lib/llvm/bin/llvm-objdump -d "$co" > "$DISASM"; fi
