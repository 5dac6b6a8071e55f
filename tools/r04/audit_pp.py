"""Dev tool: counted-wait audit of an asm-heavy kernel's ISA (the S / V phases of attention_pp.h keep LDS reads in flight across
statements).  Linear scan: every ds_read (asm or compiler) enters a FIFO with its destination registers; s_waitcnt lgkmcnt(n)
retires all but the youngest n; any instruction that names a register still in the FIFO is reported.
    python tools/r04/audit_pp.py file.s kernel_name_substring"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)$", tok)
    return {int(m.group(1))} if m else set()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n", txt, re.M):
    if pat not in m.group(1): continue
    body = txt[m.end():txt.index("s_endpgm", m.end())]
    fifo, issues, n_reads = [], [], 0
    for ln in body.split("\n"):
        t = ln.strip()
        if not t or t[0] in ";." or t.endswith(":"): continue
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(",")]
        used = set()
        for a in args:
            for tok in re.findall(r"[va]\[\d+:\d+\]|[va]\d+", a): used |= regs(tok)
        pend = set().union(*[r for r in fifo]) if fifo else set()
        if op.startswith("ds_read"):
            dst = regs(args[0])
            if pend & (used - dst): issues.append(("addr uses pending", t))
            if pend & dst: issues.append(("dest still pending", t))
            fifo.append(dst); n_reads += 1
            continue
        if op == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", t)
            if mm:
                n = int(mm.group(1)); fifo = fifo[len(fifo) - n:] if n else []
            continue
        if op in ("s_barrier",) or op.startswith("s_") or op.startswith("ds_write") and False: continue
        if pend & used: issues.append(("use before wait", t))
        if op.startswith("s_cbranch") or op == "s_branch": pass
    print(m.group(1)[:70], "ds_reads", n_reads, "issues", len(issues))
    for k, t in issues[:20]: print("   ", k, "|", t)
