"""Assemble a gfx950 .s file; branches whose target is out of the 16-bit range get a branch island (an unconditional s_branch
placed behind another unconditional branch about half way).  usage: asm_islands.py in.s out.o"""
import re, subprocess, sys
L = "/opt/rocm/lib/llvm/bin"
path, obj = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
for it in range(20):
    open(path + ".tmp.s", "w").write("\n".join(lines) + "\n")
    r = subprocess.run([L + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", path + ".tmp.s", "-o", obj], capture_output=True, text=True)
    errs = [int(m.group(1)) for m in re.finditer(r"\.tmp\.s:(\d+):\d+: error: branch size exceeds simm16", r.stderr)]
    if r.returncode == 0:
        print("assembled after", it, "rounds of islands")
        sys.exit(0)
    if not errs:
        print(r.stderr[-3000:]); sys.exit(1)
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", l)
        if m: labels[m.group(1)] = i
    uncond = [i for i, l in enumerate(lines) if l.strip().startswith("s_branch ")]
    ins = []
    for k, ln in enumerate(errs):
        i = ln - 1
        t = lines[i].split()
        tgt = t[-1]
        mid = (i + labels[tgt]) // 2
        spot = min(uncond, key=lambda u: abs(u - mid))
        name = ".Lisl_%d_%d" % (it, k)
        lines[i] = lines[i].replace(tgt, name)
        ins.append((spot, [name + ":", "\ts_branch " + tgt]))
    for spot, block in sorted(ins, reverse=True):
        lines[spot + 1:spot + 1] = block
print("gave up"); sys.exit(1)
