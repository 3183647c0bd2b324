"""Audit the ISA of the fused round kernels for the hazard the hand-counted weight ring can
create: a register that an inline-asm global_load_dwordx4 is still writing (its load has not
been released by an `s_waitcnt vmcnt`) must not be read or overwritten by any instruction.

The scan is linear over the kernel text with a FIFO model of vmcnt: every vector-memory
instruction (asm or compiler generated) enters the queue, `s_waitcnt vmcnt(N)` retires the
oldest entries until N remain (loads return in order).  Loops are seen once, in text order,
which is their first iteration; the ring keeps the same depth in steady state.  Blocks that
are only reachable by jumps start with an empty queue (optimistic), so a clean report is
necessary, not sufficient; the GPU parity tests are the other half."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "diffusionsat_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 else "kernels_round.hip")
out = "/tmp/kr_audit_%s_%d.s" % (os.path.basename(src), os.getpid())
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                       "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
text = open(out).read()


def regs(tok):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"(?<![\w\[:])v(\d+)\b", tok):
        out.add(int(a))
    return out


total_bad = 0
for m in re.finditer(r"^(_ZN4dsat\S*(?:round_p\d|chain_kernel|split_op_kernel)\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    name, lines = m.group(1), m.group(2).split("\n")
    fifo = []          # entries: set of registers being written (empty for stores / compiler loads)
    in_asm, bad, nasm, prev_op = False, [], 0, ""
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
