"""Rewrite a gfx950 assembly file (hipcc -S --cuda-device-only of vlr_kernels.hip built with -DVLR_PROFILE -DVLR_PROFILE_VALU) so
that every straight-line segment of the call kernels adds its number of VALU instructions to s100 (SCC is saved in s101 and
restored: a segment may start with SCC live).  The PROF_ADD sites of the source read s100.  usage: valu_instrument.py in.s out.s
VALU_CLASS (environment) restricts what is counted to one class of instructions, so that a run per class gives the DYNAMIC mix per
region: lane (v_readlane / v_writelane / v_readfirstlane: SGPR spill traffic and uniform reads), mov (v_mov / v_accvgpr, incl. DPP
moves), dpp (anything with a dpp / row_ / quad_perm modifier), sel (v_cndmask), cmp (v_cmp*), f64 (arithmetic on doubles), salu
(s_* instead of v_*), other (none of the above); default: every VALU instruction."""
import os, re, sys
CLASS = os.environ.get("VALU_CLASS", "all")
def counted(tj):
    op = tj.split()[0]
    if CLASS == "salu": return op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_endpgm", "s_barrier", "s_sleep", "s_setprio"))
    if not op.startswith("v_"): return False
    lane = op.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))
    mov = op.startswith(("v_mov", "v_accvgpr"))
    dpp = ("row_" in tj) or ("quad_perm" in tj) or ("wave_" in tj) or op.endswith("_dpp")
    sel = op.startswith("v_cndmask")
    cmp_ = op.startswith("v_cmp")
    f64 = ("_f64" in op) and not cmp_
    if CLASS == "all": return True
    if CLASS == "lane": return lane
    if CLASS == "mov": return mov
    if CLASS == "dpp": return dpp
    if CLASS == "sel": return sel
    if CLASS == "cmp": return cmp_
    if CLASS == "f64": return f64
    if CLASS == "other": return not (lane or mov or sel or cmp_ or f64)
    raise SystemExit("unknown VALU_CLASS " + CLASS)
src = open(sys.argv[1]).read().splitlines()
out = []
in_kernel = False
n_seg = 0
n_live = 0
def is_instr(t):
    return bool(t) and not t.startswith((".", ";", "//")) and not t.endswith(":")
def is_branch(t):
    op = t.split()[0]
    return op.startswith(("s_branch", "s_cbranch", "s_setpc", "s_swappc", "s_endpgm", "s_call", "s_trap"))
i = 0
func_re = re.compile(r"^(_ZN3vlr15vlr_call_kernelILi\d+EEEv\S*|_ZN3vlr15integrate_table\S*):")
while i < len(src):
    l = src[i]
    m = func_re.match(l)
    if m:
        in_kernel = True
    if l.startswith(".Lfunc_end"):
        in_kernel = False
    if not in_kernel:
        # descriptors of the instrumented kernels: two more SGPRs
        out.append(l)
        i += 1
        continue
    t = l.strip()
    if not is_instr(t.split(";")[0].strip()) :
        out.append(l); i += 1
        continue
    # start of a segment: collect up to and including the next branch / up to the next label / counter read
    j = i
    n = 0
    seg = []
    while j < len(src):
        tj = src[j].strip().split(";")[0].strip()
        if src[j].startswith(".Lfunc_end"): break
        if tj.endswith(":") and not tj.startswith((".loc", ".file")) and is_label(tj) if False else (re.match(r"^\.?[A-Za-z_][\w.$]*:$", tj) is not None):
            break
        seg.append(src[j])
        if is_instr(tj):
            op = tj.split()[0]
            if counted(tj): n += 1
            if is_branch(tj) or ("s100" in tj and op == "s_mov_b32"):
                j += 1
                break
        j += 1
    if n:
        # SCC at the start of the segment: dead if the segment writes it before reading it
        live = True
        for sl in seg:
            ts = sl.strip().split(";")[0].strip()
            if not is_instr(ts): continue
            op = ts.split()[0]
            if op.startswith(("s_cbranch_scc", "s_cselect", "s_addc", "s_subb", "s_cmov")): break
            if op.startswith(("s_cmp", "s_and", "s_or", "s_xor", "s_add_", "s_sub_", "s_lshl", "s_lshr", "s_ashr", "s_not", "s_bfe", "s_andn2", "s_orn2",
                              "s_nand", "s_nor", "s_xnor", "s_min", "s_max", "s_abs", "s_bcnt", "s_wqm", "s_quadmask", "s_addk", "s_cmpk", "s_bitcmp")) and "saveexec" not in op:
                live = False
                break
            if "saveexec" in op:
                live = False
                break
        if live: out.append("\ts_cselect_b32 s101, 1, 0")
        out.append("\ts_add_u32 s100, s100, %d" % n)
        if live: out.append("\ts_cmp_lg_u32 s101, 0")
        n_seg += 1
        n_live += live
    out.extend(seg)
    i = j
text = "\n".join(out) + "\n"
# kernel descriptors of the call kernels: s100, s101 are in use now
def fix(m):
    return m.group(0)
parts = text.split(".amdhsa_kernel ")
for k in range(1, len(parts)):
    if parts[k].startswith("_ZN3vlr15vlr_call_kernel"):
        parts[k] = re.sub(r"\.amdhsa_next_free_sgpr \S+", ".amdhsa_next_free_sgpr 102", parts[k], count=1)
text = ".amdhsa_kernel ".join(parts)
open(sys.argv[2], "w").write(text)
print("instrumented segments:", n_seg, "with SCC save:", n_live)
