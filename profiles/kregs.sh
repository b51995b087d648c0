#!/bin/bash
# registers / scratch / LDS of every kernel in the built engine (reads the gfx950 code object's metadata)
set -e
D=$(mktemp -d)
cp "$(dirname "$0")/../kleenexlang_amd/_build/libkxhip.so" "$D/lib.so"
(cd "$D" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1 || true)
F=$(ls "$D" | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$D/$F" | python3 -c '
import sys, re
cur = {}
for ln in sys.stdin:
    m = re.match(r"\s+\.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\S+)", ln)
    if m:
        cur[m.group(1)] = m.group(2)
    if re.match(r"\s+\.wavefront_size:", ln):
        if "name" in cur:
            print("%-4s vgpr %-4s sgpr %-5s scratch %-4s vspill  %s" % (cur.get("vgpr_count"), cur.get("sgpr_count"), cur.get("private_segment_fixed_size"), cur.get("vgpr_spill_count"), cur["name"][:110]))
        cur = {}
' | sort -k9 | c++filt 2>/dev/null || true
rm -r "$D"
