"""Audit the ISA of the fused round kernels for the hazard the hand-counted weight ring can
create: a register that an inline-asm global_load_dwordx4 is still writing (its load has not
been released by an `s_waitcnt vmcnt`) must not be read or overwritten by any instruction.

The scan is linear over the kernel text with a FIFO model of vmcnt: every vector-memory
instruction (asm or compiler generated) enters the queue, `s_waitcnt vmcnt(N)` retires the
oldest entries until N remain (loads return in order).  Loops are seen once, in text order,
which is their first iteration; the ring keeps the same depth in steady state.  Blocks that
are only reachable by jumps start with an empty queue (optimistic), so a clean report is
necessary, not sufficient; the GPU parity tests are the other half.

Second hazard (found in round 3 from a faulting probe build): gfx9 needs 5 wait states between a VALU instruction that
WRITES an SGPR (v_readlane / v_readfirstlane -- an SGPR spill reload is one -- or a VALU compare into an SGPR pair) and a
vector-memory instruction that READS that SGPR as its scalar base.  The compiler pads this for its own instructions but
does not look inside inline asm, and the ring's loads are inline asm with an SGPR base: a spill reload placed right in
front of one sends the load to a stale address (memory access fault).  Every asm load's base registers are checked
against the VALU SGPR writes of the five preceding wait states."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "diffusionsat_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 else "kernels_round.hip")
out = "/tmp/kr_audit_%s_%d.s" % (os.path.basename(src), os.getpid())
sys.path.insert(0, root)
from diffusionsat_amd.build import FLAGS   # the flags the shipped library is built with: the audited ISA is the shipped ISA
subprocess.check_call(["hipcc"] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
text = open(out).read()


def regs(tok):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"(?<![\w\[:])v(\d+)\b", tok):
        out.add(int(a))
    return out


def sregs(tok):
    out = set()
    for a, b in re.findall(r"s\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"(?<![\w\[:])s(\d+)\b", tok):
        out.add(int(a))
    return out


def valu_sgpr_writes(code):
    """SGPRs a VALU instruction writes (readlane / readfirstlane, compares and carries into SGPR pairs)."""
    op = code.split()[0]
    if not op.startswith("v_"):
        return set()
    dst = code[len(op):].split(",")[0]
    w = sregs(dst)
    if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")):
        parts = code[len(op):].split(",")
        if len(parts) > 1:
            w |= sregs(parts[1])      # the carry-out / condition pair
    return w


total_bad = 0
for m in re.finditer(r"^(_ZN4dsat\S*(?:round_p\d|chain_kernel|split_op_kernel)\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    name, lines = m.group(1), m.group(2).split("\n")
    fifo = []          # entries: set of registers being written (empty for stores / compiler loads)
    in_asm, bad, nasm, prev_op = False, [], 0, ""
    recent = []        # (wait states ago is derived from position) SGPRs written by VALU, one entry per wait state
    for ln, l in enumerate(lines):
        code = l.split(";")[0].strip()
        if "#ASMSTART" in l: in_asm = True; continue
        if "#ASMEND" in l: in_asm = False; continue
        if code.endswith(":"):
            # a block entered only by jumps (the previous instruction never falls through):
            # the queue state there is that of its predecessors, unknown to a linear scan
            if prev_op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                fifo = []
            continue
        if not code or code.startswith("."):
            continue
        w = re.match(r"s_waitcnt\s+(.*)", code)
        if w:
            prev_op = "s_waitcnt"
            v = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if v:
                n = int(v.group(1))
                while len(fifo) > n: fifo.pop(0)
            continue
        op = code.split()[0]
        prev_op = op
        if in_asm and op == "global_load_dwordx4" and ", s[" in code:
            base = sregs(code.split(",")[-1])
            stale = base & (set().union(*recent[-5:]) if recent else set())
            if stale:
                bad.append((ln, code + "   <- SGPR base written by a VALU instruction within 5 wait states", sorted(stale)))
        if op == "s_nop":
            n = int(code.split()[1], 0) if len(code.split()) > 1 else 0
            recent.extend([set()] * (n + 1))
        else:
            recent.append(valu_sgpr_writes(code))
        del recent[:-8]
        pending = set().union(*fifo) if fifo else set()
        if pending:
            touched = regs(code[len(op):])
            hit = touched & pending
            if hit and not (in_asm and op.startswith("global_load")):
                bad.append((ln, code, sorted(hit)[:4]))
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", op):
            if in_asm and op == "global_load_dwordx4":
                r = re.search(r"v\[(\d+):(\d+)\]", code)
                fifo.append(set(range(int(r.group(1)), int(r.group(2)) + 1)))
                nasm += 1
            else:
                fifo.append(set())
        if op in ("s_barrier",) and False:
            pass
    print("%s: asm loads %d, hazards %d" % (name, nasm, len(bad)))
    for ln, code, hit in bad[:12]:
        print("    line %d: %s   (pending v%s)" % (ln, code, hit))
    total_bad += len(bad)
if not os.environ.get("DSAT_AUDIT_KEEP"):
    os.remove(out)
sys.exit(1 if total_bad else 0)
