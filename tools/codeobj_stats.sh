#!/bin/bash
# Register / spill / size notes of the call kernel instances inside a built libvlr (the .hip_fatbin section is unbundled first).
#   tools/codeobj_stats.sh [path/to/libvlr.so]
L=/opt/rocm/lib/llvm/bin
SO=${1:-$(dirname "$0")/../varlociraptor_amd/libvlr.so}
W=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$SO" $W/fat.bin
python3 - "$W" <<'PY'
import sys, os
w = sys.argv[1]
d = open(os.path.join(w, "fat.bin"), "rb").read()
# concatenated clang offload bundles; split on the magic
magic = b"__CLANG_OFFLOAD_BUNDLE__"
parts = [i for i in range(len(d)) if d.startswith(magic, i)]
import struct
k = 0
for st in parts:
    n = struct.unpack_from("<Q", d, st + 24)[0]
    off = st + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", d, off)
        trip = d[off + 24: off + 24 + tl].decode()
        off += 24 + tl
        if "gfx" in trip and sz:
            open(os.path.join(w, "co%d.elf" % k), "wb").write(d[st + o: st + o + sz]); k += 1
PY
for f in $W/co*.elf; do
  $L/llvm-readelf --notes $f 2>/dev/null | python3 -c "
import sys, re
t = sys.stdin.read()
for blk in re.split(r'\n\s+- (?=\.a)', t):   # one block per kernel (its keys are sorted: the block starts at .agpr_count / .args)
    nm = re.search(r'\.name:\s+(\S+)', blk)
    if not nm or ('vlr_call_kernel' not in nm.group(1) and 'afd_kernel' not in nm.group(1)): continue
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [0, '?'])[1]
    print(nm.group(1)[:60], 'vgpr', g('vgpr_count'), 'vspill', g('vgpr_spill_count'), 'sgpr', g('sgpr_count'), 'sspill', g('sgpr_spill_count'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
"
  ls -l $f | awk "{print \"code object bytes\", \$5}"
done
rm -rf $W
