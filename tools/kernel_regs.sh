#!/bin/bash
# Register / spill / scratch metadata of every kernel in a built librtmi.so (read from the embedded code object).
# Usage: tools/kernel_regs.sh [path/to/librtmi.so]
SO=$(realpath "${1:-$(dirname "$0")/../ray-tracing-cuda_amd/lib/librtmi.so}")
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cd "$TMP"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input="$SO" >/dev/null 2>&1
# the fat binary sits in .hip_fatbin; extract the gfx950 code object
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$SO" fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=co.o 2>/dev/null || { echo "unbundle failed"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes co.o | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    print("%-60s vgpr %4s sgpr %4s vgpr_spill %4s sgpr_spill %4s scratch %5s lds %6s" % (
        name[:60], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
