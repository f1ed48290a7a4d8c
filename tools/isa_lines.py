"""Static VALU instruction counts of one kernel per source function / source line, from `hipcc -S -gline-tables-only`.
   tools/isa_lines.py <file.s> <kernel symbol substring> [line-level function name]"""
import re, sys, collections, subprocess
asm, ksub = sys.argv[1], sys.argv[2]
detail = sys.argv[3] if len(sys.argv) > 3 else None
src = open("varlociraptor_amd/csrc/vlr_kernels.hip").read().splitlines()
# function start lines of the source (top-level definitions)
starts = []
for i, l in enumerate(src, 1):
    m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|static|inline)[^;]*?\b(\w+)\s*\(", l)
    if m and not l.strip().endswith(";"):
        starts.append((i, m.group(1)))
    elif re.match(r"^__global__", l):
        m2 = re.search(r"(\w+)\s*\(", l)
        if m2: starts.append((i, m2.group(1)))
def fn_of(line):
    name = "?"
    for s, n in starts:
        if s <= line: name = n
        else: break
    return name
text = open(asm).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(ksub), text, re.M)
beg = m.start(); end = text.index(".Lfunc_end", beg)
cur = (0, 0)
per_fn = collections.defaultdict(collections.Counter)
per_line = collections.defaultdict(collections.Counter)
def cls(t):
    op = t.split()[0]
    if op in ("v_readlane_b32", "v_writelane_b32"): return "spill"
    if op == "v_readfirstlane_b32": return "rfl"
    if op.startswith("v_"):
        if "dpp" in t or "row_" in t or "quad_perm" in t: return "dpp"
        if "f64" in op and not op.startswith("v_cmp"): return "f64"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_cmp"): return "cmp"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
        return "vint"
    if op.startswith("s_"): return "s"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem"
    return "oth"
for l in text[beg:end].splitlines():
    t = l.strip()
    if t.startswith(".loc"):
        p = t.split()
        cur = (int(p[1]), int(p[2]))
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    t = t.split(";")[0].strip()
    if not t: continue
    c = cls(t)
    f = fn_of(cur[1]) if cur[0] == 0 else "file%d" % cur[0]
    per_fn[f][c] += 1
    if detail and f == detail: per_line[cur[1]][c] += 1
V = ("f64", "cnd", "cmp", "mov", "vint", "dpp", "spill", "rfl")
rows = []
for f, c in per_fn.items():
    rows.append((sum(c[k] for k in V), f, c))
rows.sort(reverse=True)
print("%-28s %6s | %s | %5s %5s" % ("function", "VALU", " ".join("%5s" % k for k in V), "salu", "lds"))
tot = collections.Counter()
for n, f, c in rows:
    tot.update(c)
    if n >= 15: print("%-28s %6d | %s | %5d %5d" % (f, n, " ".join("%5d" % c[k] for k in V), c["s"], c["lds"]))
print("%-28s %6d | %s | %5d %5d" % ("TOTAL", sum(tot[k] for k in V), " ".join("%5d" % tot[k] for k in V), tot["s"], tot["lds"]))
if detail:
    for ln in sorted(per_line):
        c = per_line[ln]
        n = sum(c[k] for k in V)
        if n: print("%5d %4d | %s | %s" % (ln, n, " ".join("%3d" % c[k] for k in V), src[ln - 1].strip()[:110]))
