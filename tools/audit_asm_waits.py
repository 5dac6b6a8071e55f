"""Dev tool: audit the library's gfx950 ISA for the one thing hipcc cannot know about hand-written LDS reads -- that the destination
registers of an asm `ds_read` belong to the hardware until an `s_waitcnt lgkmcnt` has passed.  hipcc may place copies of an asm
statement's outputs anywhere behind it; where the wait is a LATER asm statement such a copy reads registers the data has not reached
yet (found once, in the split GEMM's epilogue, as wrong results in 1.4 % of the outputs).  The kernels therefore keep read + wait in
one statement wherever the result is consumed by compiler code; this script checks the places that do not.

    python tools/audit_asm_waits.py            # compiles medtok_vq.hip to assembly (~2 min) and scans every kernel

Linear scan per kernel in layout order: registers written by an asm ds_read are 'pending' until an s_waitcnt lgkmcnt (asm or
compiler-inserted); any compiler instruction that names a pending register is reported.
Second check: a VALU write to the data registers of a 96/128-bit LDS store needs two wait states on gfx950, which hipcc counts for
its own stores only (found as lanes 8-15, 24-31, ... storing the next quad's .xy): every asm ds_write_b96/b128 is followed for
two wait states and a VALU instruction that writes its data registers inside them is reported; so is a VALU write to those
registers by the instruction directly in front of the store (one wait state in that direction)."""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(asm_text):
    total = 0
    for k in re.findall(r"^(_Z[\w]+):[^\n]*\n", asm_text, re.M):      # (the label line may carry a comment)
        i = asm_text.find("\n" + k + ":") + 1
        j = asm_text.find("s_endpgm", i)
        pending, in_asm, issues = set(), False, []
        dead_end = False                                 # the last instruction was an unconditional branch: the next label starts afresh
        body = []                                        # (instruction, inside an asm statement) in layout order, for the store checks
        for ln in asm_text[i:j].split("\n"):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if t.endswith(":") and dead_end:             # a block only reached by jumps: what was pending on the fall-through path is not here
                pending.clear()
                body.append(("s_nop 15", False))         # (and the store checks do not look across it)
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            op = t.split()[0]
            dead_end = op == "s_branch"
            body.append((t, in_asm))
            args = [a.strip() for a in t[len(op):].split(",")]
            if in_asm:
                if op.startswith("ds_read"):
                    pending |= regs_of(args[0])
                elif op == "s_waitcnt" and "lgkmcnt" in t:
                    pending.clear()                      # (counted waits inside a statement: the author's arithmetic)
                continue
            if op == "s_waitcnt" and "lgkmcnt(0)" in t:
                pending.clear()
                continue
            used = set()
            for a in args:
                for tok in re.findall(r"v\[\d+:\d+\]|v\d+", a):
                    used |= regs_of(tok)
            if pending & used:
                issues.append(t)
        # wide LDS stores FROM ASM STATEMENTS (hipcc looks after its own): two wait states before a VALU write to their data
        # registers, one behind a VALU write to them
        for n, (b, b_asm) in enumerate(body):
            if not b_asm or not (b.startswith("ds_write_b128") or b.startswith("ds_write_b96")):
                continue
            data = regs_of(b.split(",")[1].strip().split()[0]) if "," in b else set()
            states = 0
            for nxt, _ in body[n + 1:n + 4]:
                if states >= 2:
                    break
                op = nxt.split()[0]
                if op == "s_nop":
                    states += int(nxt.split()[1]) + 1
                    continue
                if op.startswith("v_") and regs_of(nxt[len(op):].split(",")[0].strip()) & data:
                    issues.append(b + "   <-   " + nxt)
                states += 1
            if n > 0:
                prv = body[n - 1][0]
                op = prv.split()[0]
                if op.startswith("v_") and not op.startswith("v_cmp") and regs_of(prv[len(op):].split(",")[0].strip()) & data:
                    issues.append(prv + "   ->   " + b)
        if issues:
            total += len(issues)
            print(k[:90], len(issues))
            for x in issues[:8]:
                print("      ", x)
    return total


def audit_mfma_asm_readers(asm_text):
    """Third check: an XDL (MFMA) result needs passes + 2 wait states before a VALU instruction may read it; hipcc's hazard recognizer
    counts them for its own VALU instructions, not for inline-asm readers (found as limits computed from half-finished sums in
    filter_rows64_kernel).  Per kernel, in layout order: behind every v_mfma its destination registers are 'cooking' for passes + 2
    wait states (16 passes for 32x32x16, 8 for 16x16x32; an instruction is one wait state, s_nop N is N + 1); an instruction INSIDE
    an asm statement that reads a cooking register (other than another MFMA: those interlock) is reported."""
    total = 0
    for k in re.findall(r"^(_Z[\w]+):[^\n]*\n", asm_text, re.M):      # (the label line may carry a comment)
        i = asm_text.find("\n" + k + ":") + 1
        j = asm_text.find("s_endpgm", i)
        cooking, in_asm, issues = [], False, []          # [registers, wait states left]
        for ln in asm_text[i:j].split("\n"):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            op = t.split()[0]
            args = [a.strip() for a in t[len(op):].split(",")]
            states = (int(args[0]) + 1) if op == "s_nop" else 1
            if in_asm and op.startswith("v_") and not op.startswith("v_mfma"):
                used = set()
                for a in args[1:]:
                    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", a):
                        used |= regs_of(tok)
                for regs, left in cooking:
                    if left > 0 and regs & used:
                        issues.append(f"{t}   (an MFMA result with {left} wait states to go)")
                        break
            cooking = [(r, left - states) for r, left in cooking if left - states > 0]
            if op.startswith("v_mfma"):
                passes = 16 if "32x32" in op else 8
                cooking.append((regs_of(args[0]), passes + 2))
        if issues:
            total += len(issues)
            print(k[:90], len(issues))
            for x in issues[:6]:
                print("      ", x)
    return total


if __name__ == "__main__":
    if len(sys.argv) > 1:
        text = Path(sys.argv[1]).read_text()
    else:
        # the SHIPPED flags (medtok_amd/csrc/build.py), with the link step swapped for an assembly listing: if the build's flags
        # change, the audited ISA changes with them
        sys.path.insert(0, str(ROOT))
        from medtok_amd.csrc import build as hip_build
        try:
            cc = hip_build.hipcc()
        except RuntimeError as exc:
            print("SKIP:", exc)
            sys.exit(77)
        flags = [f for f in hip_build.FLAGS if f not in ("-shared", "-fPIC")]
        with tempfile.TemporaryDirectory() as tmp:
            out = Path(tmp) / "vq.s"
            subprocess.check_call([cc, *flags, "-S", "--cuda-device-only", "-o", str(out), str(hip_build.SRC)])
            text = out.read_text()
    print("kernels scanned:", len(re.findall(r"^(_Z[\w]+):[^\n]*\n", text, re.M)))
    n = audit(text)
    print("uses of an asm ds_read's destination before a wait / VALU writes into a wide asm LDS store's data within two wait states:", n)
    m = audit_mfma_asm_readers(text)
    print("asm VALU readers of an MFMA result inside its wait states:", m)
    sys.exit(1 if n or m else 0)
