"""Static check of a gfx950 assembly listing (hipcc -S --cuda-device-only) for the hazard inline-asm vector loads are
exposed to: a VALU instruction writing an SGPR (v_readlane / v_readfirstlane, e.g. an SGPR spill reload) within five
wait states of a global_load that uses that SGPR as its scalar base.  The compiler pads its own instructions, not the
ones inside inline asm.  Probes are checked with this before they run on a GPU.
    python tools/asm_sgpr_hazard.py file.s"""
import re, sys
def sregs(tok):
    out = set()
    for a, b in re.findall(r"s\[(\d+):(\d+)\]", tok): out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"(?<![\w\[:])s(\d+)\b", tok): out.add(int(a))
    return out
text = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r"^(\S+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    recent, in_asm, n = [], False, 0
    for l in m.group(2).split("\n"):
        if "#ASMSTART" in l: in_asm = True; continue
        if "#ASMEND" in l: in_asm = False; continue
        c = l.split(";")[0].strip()
        if not c or c.startswith(".") or c.endswith(":"): continue
        op = c.split()[0]
        if in_asm and op.startswith("global_load") and ", s[" in c:
            n += 1
            st = sregs(c.split(",")[-1]) & (set().union(*recent[-5:]) if recent else set())
            if st: bad += 1; print("%s: %s   <- base s%s written by VALU within 5 wait states" % (m.group(1), c, sorted(st)))
        if op == "s_nop": recent.extend([set()] * (int(c.split()[1], 0) + 1))
        elif op.startswith(("v_readlane", "v_readfirstlane")) or (op.startswith("v_cmp") and "s[" in c.split(",")[0]):
            recent.append(sregs(c[len(op):].split(",")[0]))
        else: recent.append(set())
        del recent[:-8]
    if n: print("%s: %d asm loads" % (m.group(1), n))
print("hazards", bad)
sys.exit(1 if bad else 0)
