"""Static view of a kernel's ISA: natural loops (backward branches) with their instruction mix.
   tools/isa_loops.py <file.s> [min_instructions]     (file = one kernel's slice of `hipcc -S --cuda-device-only`)"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
minsz = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lab = {}
ins = []   # (lineno, text)
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    ins.append((i + 1, t.split(";")[0].strip()))
def cls(t):
    op = t.split()[0]
    if op in ("v_readlane_b32", "v_writelane_b32"): return "spill"
    if op == "v_readfirstlane_b32": return "rfl"
    if op.startswith("v_"):
        if "dpp" in t or "row_" in t or "quad_perm" in t: return "dpp"
        if "f64" in op: return "f64"
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem"
    return "other"
loops = []
for k, (ln, t) in enumerate(ins):
    if t.startswith(("s_cbranch", "s_branch")):
        tgt = t.split()[-1]
        if tgt in lab and lab[tgt] <= k:
            loops.append((lab[tgt], k))
loops.sort(key=lambda x: (x[0], -x[1]))
print("%d instructions, %d backward branches" % (len(ins), len(loops)))
for a, b in loops:
    n = b - a + 1
    if n < minsz: continue
    c = {}
    for _, t in ins[a:b + 1]:
        c[cls(t)] = c.get(cls(t), 0) + 1
    depth = sum(1 for (x, y) in loops if x <= a and y >= b) - 1
    print("%s lines %6d-%6d  n=%5d  " % ("  " * depth, ins[a][0], ins[b][0], n) + " ".join("%s=%d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))
