"""Audit the ISA of the fused round kernels: registers written by the inline-asm weight loads
(global_load_dwordx4 inside ;;#ASMSTART/;;#ASMEND) must never be the source of a compiler
v_mov / scratch store between the first and the last such load (a copy of a register whose
load may still be in flight reads garbage)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "diffusionsat_amd", "csrc", "kernels_round.hip")
out = "/tmp/kr_audit.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                       "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
text = open(out).read()
bad = 0
for m in re.finditer(r"^(_ZN4dsat\S*round_p\d\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, lines = m.group(1), m.group(2).split("\n")
    ring, idx, in_asm = set(), [], False
    for i, l in enumerate(lines):
        if "#ASMSTART" in l: in_asm = True
        elif "#ASMEND" in l: in_asm = False
        elif in_asm:
            r = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
            if r:
                ring.update(range(int(r.group(1)), int(r.group(2)) + 1))
                idx.append(i)
    hits, in_asm = [], False
    for l in lines[idx[0]:idx[-1] + 1] if idx else []:
        if "#ASMSTART" in l: in_asm = True
        elif "#ASMEND" in l: in_asm = False
        if in_asm: continue
        mm = re.match(r"\s*(v_mov_b32_e32|v_mov_b64_e32|scratch_store_\w+|v_accvgpr_write_b32)\s+(.*)", l)
        if not mm: continue
        ops = [o.strip() for o in mm.group(2).split(";")[0].split(",")]
        srcs = ops[1:2] if mm.group(1).startswith("scratch") else ops[1:]
        for sreg in srcs:
            r = re.match(r"v\[(\d+):(\d+)\]$|v(\d+)$", sreg)
            if r:
                regs = range(int(r.group(1)), int(r.group(2)) + 1) if r.group(1) else [int(r.group(3))]
                if any(x in ring for x in regs): hits.append(l.strip())
    print("%-62s asm loads %3d  ring regs %3d  compiler copies of ring regs: %d" % (name[:62], len(idx), len(ring), len(hits)))
    for h in hits[:8]: print("      ", h)
    bad += len(hits)
sys.exit(1 if bad else 0)
