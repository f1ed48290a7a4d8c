"""Static EXEC audit of a gfx950 kernel's ISA (VERDICT r05 "next" #1b).

For one kernel of an assembly listing (`hipcc -S --cuda-device-only -gline-tables-only`) this computes, by a forward must-dataflow
over the control-flow graph, at which instructions EXEC is PROVABLY the launch mask (all 64 lanes: the workgroup is one full
wave), and reports

  A. every cross-lane read that executes where EXEC may be partial: DPP operands (with bound_ctrl a disabled source lane reads
     as 0, without it the destination keeps its old value), ds_bpermute / ds_swizzle (disabled lanes do not take part), user
     v_readlane (ignores EXEC, listed for its source) — SGPR-spill reloads (v_readlane of a slot last written by v_writelane)
     are not cross-lane reads of program values and are only counted;
  B. every cross-lane read under full EXEC one of whose source VGPRs has a reaching definition made under partial EXEC
     (such a definition leaves the lanes that were disabled with whatever the register held before: legal when an earlier
     full definition of the SAME value reaches too — `v = 0; if (c) v = x;` —, a defect when the register held another value).

How "EXEC is full" is proven: the kernel starts with it; `s_and_saveexec_b64 sX` under full EXEC leaves the full mask in sX;
`s_or_b64 exec, exec, sX` / `s_mov_b64 exec, sX` with such an sX (also through SGPR copies and v_writelane / v_readlane spill
slots) restores it; every other write of EXEC makes it unknown; facts are intersected where paths meet.  The masks that end
divergent LOOPS (`s_or_b64 exec, exec, <lanes that left>`) cannot be followed that way, so one structural rule is added: a block
that post-dominates its immediate dominator runs under the EXEC its dominator was entered with (what the structurised control
flow of this compiler guarantees: every region restores EXEC at its exit) — the audit checks where the PROGRAM puts cross-lane
reads, not the compiler's mask bookkeeping.

usage: python tools/isa_exec_audit.py file.s <kernel-symbol-substring> [--list-a] [--list-b] [--ctx N]
"""
import re
import sys
from collections import defaultdict

CROSS_DPP = re.compile(r"quad_perm|row_shl|row_shr|row_ror|row_mirror|row_half_mirror|row_bcast|row_newbcast|row_share|row_xmask|wave_sh|wave_ro")


def parse_kernel(path, sym):
    lines = open(path).read().splitlines()
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^[A-Za-z_][\w.$]*:", l) and sym in l.split(":")[0] and not l.startswith(".L"):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        raise SystemExit("kernel %r not found" % sym)
    ins = []      # dicts: op, args(list of str), text, src (source line), asm (listing line)
    labels = {}
    src = 0
    via = ()
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.split(";")[0].strip()
        if not t:
            continue
        if t.startswith(".loc"):   # innermost line; the inlining chain is in the comment: "file:LINE:COL @[ file:LINE:COL @[ ... ] ]"
            p = t.split()
            src = int(p[2])
            chain = re.findall(r":(\d+):\d+", l.split(";", 1)[1]) if ";" in l else []
            via = tuple(int(x) for x in chain[1:])
            continue
        if t.startswith(".") or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        args = [a.strip() for a in re.split(r",(?![^\[]*\])", rest.strip())] if rest.strip() else []
        ins.append({"op": op, "args": args, "text": t, "src": src, "via": via, "asm": i + 1})
    return ins, labels


def regs_of(arg):
    """32-bit register names an operand covers: 's4', 's[4:5]', 'v[10:11]', 'vcc', 'exec' ..."""
    a = arg.split()[0] if arg else ""
    a = a.strip("|").lstrip("-")
    m = re.match(r"^([sv])\[(\d+):(\d+)\]$", a)
    if m:
        return ["%s%d" % (m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"^([sv])(\d+)$", a)
    if m:
        return [a]
    if a == "vcc":
        return ["vcc_lo", "vcc_hi"]
    if a == "exec":
        return ["exec_lo", "exec_hi"]
    if a in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0", "scc"):
        return [a]
    return []


NO_DEST = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep",
           "s_setreg", "s_sendmsg", "s_icache", "s_dcache", "s_trap", "s_sethalt", "s_set_gpr", "s_code_end", "s_memtime_dummy",
           "ds_write", "ds_store", "global_store", "scratch_store", "buffer_store", "flat_store", "global_atomic", "s_store",
           "s_wakeup", "s_ttrace", "s_decperf", "s_incperf")


def dests(i):
    op, args = i["op"], i["args"]
    if op.startswith(NO_DEST) or not args:
        if op.startswith("global_atomic") and len(args) == 4:  # returning atomic: vdst first
            return regs_of(args[0])
        return []
    d = regs_of(args[0])
    # carry-out / second destinations
    if op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")) and len(args) > 1:
        d = d + regs_of(args[1])
    if op.startswith("v_cmp") and not d:  # e32 compares write vcc (v_cmpx: exec)
        d = ["vcc_lo", "vcc_hi"]
    if op.startswith("v_cmpx"):
        d = d + ["exec_lo", "exec_hi"]
    if op in ("s_swappc_b64", "s_getpc_b64", "s_call_b64"):
        pass
    if op.startswith("ds_read2") or op.startswith("ds_load2"):
        pass
    return d


def sources(i):
    op, args = i["op"], i["args"]
    if op.startswith(NO_DEST):
        return [r for a in args for r in regs_of(a)]
    out = []
    for k, a in enumerate(args):
        if k == 0:
            continue
        out += regs_of(a)
    if op == "v_writelane_b32":  # read-modify-write of the destination VGPR
        out += regs_of(args[0])
    return out


def is_dpp(i):
    return bool(CROSS_DPP.search(i["text"]))


class State:
    __slots__ = ("full", "tags")

    def __init__(self, full, tags):
        self.full = full      # EXEC is the launch mask
        self.tags = tags      # name -> 'FL' | 'FH' | 'SP'   (names: sN, vcc_lo/hi, slot 'vN@L')

    def copy(self):
        return State(self.full, dict(self.tags))

    def meet(self, o):
        ch = False
        if self.full and not o.full:
            self.full = False
            ch = True
        for k in list(self.tags):
            v = o.tags.get(k)
            if v != self.tags[k]:
                if v is not None and v != "SP" and self.tags[k] != "SP" or v is None:
                    del self.tags[k]
                else:  # both spill slots, one of them a full-mask half: keep as plain spill
                    if self.tags[k] != "SP":
                        self.tags[k] = "SP"
                    else:
                        continue
                ch = True
        return ch


def pair_full(st, arg):
    r = regs_of(arg)
    if arg.strip() == "-1":
        return True
    return len(r) == 2 and st.tags.get(r[0]) == "FL" and st.tags.get(r[1]) == "FH"


def transfer(st, i):
    op, args = i["op"], i["args"]
    t = st.tags
    if op == "v_writelane_b32":
        v = regs_of(args[0])[0]
        s = regs_of(args[1])
        lane = args[2]
        tag = t.get(s[0]) if s else None
        t["%s@%s" % (v, lane)] = tag if tag in ("FL", "FH") else "SP"
        return
    if op == "v_readlane_b32":
        d = regs_of(args[0])
        v = regs_of(args[1])[0]
        tag = t.get("%s@%s" % (v, args[2]))
        for r in d:
            t.pop(r, None)
        if tag in ("FL", "FH") and d:
            t[d[0]] = tag
        return
    old_full = st.full
    if op in ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64", "s_xor_saveexec_b64", "s_orn2_saveexec_b64", "s_nand_saveexec_b64", "s_nor_saveexec_b64", "s_xnor_saveexec_b64"):
        srcfull = pair_full(st, args[1])
        d = regs_of(args[0])
        for r in d:
            t.pop(r, None)
        if old_full and len(d) == 2:
            t[d[0]] = "FL"
            t[d[1]] = "FH"
        st.full = (op == "s_or_saveexec_b64" and (old_full or srcfull)) or (op == "s_and_saveexec_b64" and old_full and srcfull)
        return
    d = dests(i)
    if "exec_lo" in d or "exec_hi" in d:
        if op == "s_or_b64" and any(pair_full(st, a) for a in args[1:]):
            st.full = True
        elif op == "s_or_b64" and old_full:
            st.full = True
        elif op == "s_mov_b64":
            st.full = pair_full(st, args[1])
        else:
            st.full = False
        return
    if op == "s_mov_b64" and len(args) == 2:
        dd = regs_of(args[0])
        if args[1].strip() == "exec":
            for r in dd:
                t.pop(r, None)
            if old_full and len(dd) == 2:
                t[dd[0]] = "FL"
                t[dd[1]] = "FH"
            return
        ss = regs_of(args[1])
        tags = [t.get(r) for r in ss]
        for r in dd:
            t.pop(r, None)
        if len(ss) == len(dd):
            for r, g in zip(dd, tags):
                if g in ("FL", "FH"):
                    t[r] = g
        return
    if op == "s_mov_b32" and len(args) == 2:
        dd = regs_of(args[0])
        ss = regs_of(args[1])
        g = t.get(ss[0]) if ss else None
        for r in dd:
            t.pop(r, None)
        if g in ("FL", "FH") and dd:
            t[dd[0]] = g
        return
    for r in d:
        t.pop(r, None)
        if r[0] == "v":  # a VALU / memory write of a spill VGPR invalidates its slots
            pre = r + "@"
            for k in [k for k in t if k.startswith(pre)]:
                del t[k]


def successors(ins, labels, k):
    i = ins[k]
    op = i["op"]
    if op == "s_endpgm":
        return []
    if op == "s_branch":
        return [labels[i["args"][0]]]
    if op.startswith("s_cbranch"):
        s = [labels[i["args"][-1]]]
        if k + 1 < len(ins):
            s.append(k + 1)
        return s
    if op == "s_setpc_b64":
        return []
    return [k + 1] if k + 1 < len(ins) else []


def idoms(nblk, succ, pred, entry):
    """immediate dominators (Cooper / Harvey / Kennedy) of the blocks reachable from entry; unreachable: None"""
    order, seen, stack = [], {entry}, [(entry, iter(succ[entry]))]
    while stack:
        j, it = stack[-1]
        for s_ in it:
            if s_ not in seen:
                seen.add(s_)
                stack.append((s_, iter(succ[s_])))
                break
        else:
            order.append(j)
            stack.pop()
    rpo = order[::-1]
    num = {b: k for k, b in enumerate(rpo)}
    idom = [None] * nblk
    idom[entry] = entry
    changed = True
    while changed:
        changed = False
        for b in rpo[1:]:
            new = None
            for p_ in pred[b]:
                if idom[p_] is None or p_ not in num:
                    continue
                if new is None:
                    new = p_
                    continue
                a, c = p_, new
                while a != c:
                    while num[a] > num[c]:
                        a = idom[a]
                    while num[c] > num[a]:
                        c = idom[c]
                new = a
            if new is not None and idom[b] != new:
                idom[b] = new
                changed = True
    return idom


def main():
    path, sym = sys.argv[1], sys.argv[2]
    list_a = "--list-a" in sys.argv
    list_b = "--list-b" in sys.argv
    ins, labels = parse_kernel(path, sym)
    n = len(ins)
    # basic blocks
    leaders = {0} | set(labels.values())
    for k in range(n):
        if ins[k]["op"].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) and k + 1 < n:
            leaders.add(k + 1)
    leaders = sorted(leaders)
    bstart = {k: j for j, k in enumerate(leaders)}
    bend = [leaders[j + 1] if j + 1 < len(leaders) else n for j in range(len(leaders))]
    succ = [[bstart[s] for s in successors(ins, labels, bend[j] - 1)] for j in range(len(leaders))]
    pred = defaultdict(list)
    for j, ss in enumerate(succ):
        for s in ss:
            pred[s].append(j)
    # dominators, post-dominators (virtual exit behind every block without successors)
    nb_ = len(leaders)
    idom = idoms(nb_, succ, pred, 0)
    rsucc = [list(pred[j]) for j in range(nb_)] + [[j for j in range(nb_) if not succ[j] and idom[j] is not None]]
    rpred = defaultdict(list)
    for j, ss in enumerate(rsucc):
        for s_ in ss:
            rpred[s_].append(j)
    ipdom = idoms(nb_ + 1, rsucc, rpred, nb_)

    def postdominates(b, d):  # b post-dominates d
        x = d
        while x is not None and x != nb_:
            if x == b:
                return True
            nx = ipdom[x]
            if nx == x:
                break
            x = nx
        return False

    structural = [False] * nb_   # block runs under the EXEC its immediate dominator was entered with
    for j in range(1, nb_):
        if idom[j] is not None and idom[j] != j and postdominates(j, idom[j]):
            structural[j] = True
    # forward must-dataflow
    IN = [None] * len(leaders)
    IN[0] = State(True, {})
    work = [0]
    while work:
        j = work.pop()
        st = IN[j].copy()
        for k in range(leaders[j], bend[j]):
            transfer(st, ins[k])
        for s in succ[j]:
            if IN[s] is None:
                IN[s] = st.copy()
                work.append(s)
            elif IN[s].meet(st):
                work.append(s)
    # structural rule, to a fixed point with the dataflow (fullness only ever grows here; tags stay as computed: conservative)
    changed = True
    while changed:
        changed = False
        for j in range(1, nb_):
            if IN[j] is None or IN[j].full or not structural[j]:
                continue
            d = idom[j]
            if IN[d] is not None and IN[d].full:
                IN[j].full = True
                changed = True
                # push forward through successors whose other predecessors agree
                work = [j]
                while work:
                    b = work.pop()
                    st = IN[b].copy()
                    for k in range(leaders[b], bend[b]):
                        transfer(st, ins[k])
                    for s_ in succ[b]:
                        if IN[s_] is None or IN[s_].full or not st.full:
                            continue
                        ok = True
                        for p_ in pred[s_]:
                            if IN[p_] is None:
                                continue
                            sp = IN[p_].copy()
                            for k in range(leaders[p_], bend[p_]):
                                transfer(sp, ins[k])
                            if not sp.full:
                                ok = False
                                break
                        if ok:
                            IN[s_].full = True
                            work.append(s_)
    # per-instruction facts
    full_at = [False] * n
    spill_reload = [False] * n
    for j in range(len(leaders)):
        if IN[j] is None:
            continue
        st = IN[j].copy()
        for k in range(leaders[j], bend[j]):
            i = ins[k]
            full_at[k] = st.full
            if i["op"] == "v_readlane_b32":
                v = regs_of(i["args"][1])[0]
                spill_reload[k] = ("%s@%s" % (v, i["args"][2])) in st.tags
            transfer(st, i)
    reach = sum(1 for j in range(len(leaders)) if IN[j] is not None)
    print("%s: %d instructions, %d blocks (%d reachable), EXEC provably full at %d instructions (%.1f %%)" %
          (sym, n, len(leaders), reach, sum(full_at), 100.0 * sum(full_at) / n))
    kinds = defaultdict(lambda: [0, 0])
    siteA = defaultdict(list)
    cross = []
    for k, i in enumerate(ins):
        op = i["op"]
        kind = None
        if is_dpp(i):
            kind = "dpp"
        elif op == "v_readlane_b32":
            kind = "spill_reload" if spill_reload[k] else "readlane"
        elif op == "v_writelane_b32":
            kind = "writelane"
        elif op == "v_readfirstlane_b32":
            kind = "readfirstlane"
        elif op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
            kind = op.split("_b32")[0]
        elif op.startswith("v_permlane"):
            kind = "permlane"
        if kind is None:
            continue
        kinds[kind][0] += 1
        if not full_at[k]:
            kinds[kind][1] += 1
            if kind not in ("spill_reload", "writelane", "readfirstlane"):
                siteA[(kind, i["src"])].append(k)
        if kind in ("dpp", "readlane", "ds_bpermute", "ds_permute", "ds_swizzle", "permlane"):
            cross.append((k, kind))
    print("\ncross-lane instructions: total / where EXEC is not provably full")
    for kind, (a, b) in sorted(kinds.items()):
        print("  %-14s %6d %6d" % (kind, a, b))
    print("\nA. cross-lane reads of program values under possibly partial EXEC, by source line: %d lines, %d instructions" %
          (len(siteA), sum(len(v) for v in siteA.values())))
    for (kind, src), ks in sorted(siteA.items(), key=lambda kv: kv[0][1]):
        vias = sorted(set(ins[k]["via"] for k in ks))
        print("  src line %5d  %-12s x%d   e.g. asm %d: %s" % (src, kind, len(ks), ins[ks[0]]["asm"], ins[ks[0]]["text"]))
        for v in vias[:12]:
            print("        inlined at lines %s" % (" <- ".join(str(x) for x in v) if v else "(top level)"))
        if list_a:
            for k in ks:
                print("        asm %d: %s" % (ins[k]["asm"], ins[k]["text"]))
    # B: reaching definitions under partial EXEC for cross-lane reads under full EXEC
    blk_of = [0] * n
    for j in range(len(leaders)):
        for k in range(leaders[j], bend[j]):
            blk_of[k] = j
    defs_in_block = [defaultdict(list) for _ in leaders]   # reg -> [k...] ascending
    for k, i in enumerate(ins):
        for r in dests(i):
            if r[0] == "v":
                defs_in_block[blk_of[k]][r].append(k)

    def reaching_partial(k, reg):
        """definitions of reg made under partial EXEC that reach instruction k along some path without an intervening full-EXEC def"""
        found = []
        seen = set()
        stack = [(blk_of[k], k)]
        while stack:
            j, upto = stack.pop()
            ds = [d for d in defs_in_block[j].get(reg, []) if d < upto]
            stop = False
            for d in reversed(ds):
                if ins[d]["op"] == "v_writelane_b32":
                    continue
                if full_at[d]:
                    stop = True
                    break
                found.append(d)
            if stop:
                continue
            for pj in pred[j]:
                if pj not in seen:
                    seen.add(pj)
                    stack.append((pj, bend[pj]))
        return found

    nb = 0
    siteB = defaultdict(list)
    for k, kind in cross:
        i = ins[k]
        if not full_at[k]:
            continue
        srcs = [r for r in sources(i) if r[0] == "v"]
        if kind == "dpp":  # only the permuted operand (src0) is read across lanes; `old` (the destination) matters without bound_ctrl
            srcs = [r for r in regs_of(i["args"][1])] if len(i["args"]) > 1 else []
        elif kind == "readlane":
            srcs = regs_of(i["args"][1])
        for r in srcs:
            pd = reaching_partial(k, r)
            if pd:
                nb += 1
                siteB[(kind, i["src"])].append((k, r, pd))
    print("\nB. cross-lane reads under full EXEC with a source defined under partial EXEC: %d operand uses on %d source lines" % (nb, len(siteB)))
    for (kind, src), lst in sorted(siteB.items(), key=lambda kv: kv[0][1]):
        k, r, pd = lst[0]
        print("  src line %5d  %-10s x%d   e.g. asm %d: %s   <- %s defined at asm %s (src %s)" %
              (src, kind, len(lst), ins[k]["asm"], ins[k]["text"], r, ",".join(str(ins[d]["asm"]) for d in pd[:4]), ",".join(str(ins[d]["src"]) for d in pd[:4])))
        if list_b:
            for k, r, pd in lst:
                print("        asm %d: %s   %s <- %s" % (ins[k]["asm"], ins[k]["text"], r, ["%d:%s" % (ins[d]["asm"], ins[d]["text"]) for d in pd[:6]]))


if __name__ == "__main__":
    main()
