"""The hand-counted weight ring (fused_common.h) relies on a property the compiler does not
know about: a register an inline-asm load is still writing must not be touched before the
`s_waitcnt` that releases it.  tools/audit_ring.py replays a FIFO model of vmcnt over the
gfx950 ISA of the fused kernels; this test keeps it clean (hipcc cross-compiles here)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("src", ["kernels_round.hip", "kernels_chain.hip", "kernels_split_op.hip"])
def test_no_instruction_touches_a_register_with_a_load_in_flight(src):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_ring.py"), src],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert "asm loads" in res.stdout and "hazards 0" in res.stdout


def test_round_kernels_spill_nothing_and_pad_their_asm_stores(tmp_path):
    """Two properties of the gfx950 ISA of kernels_round.hip that cost dearly when lost and that no
    functional test of a small batch notices:
      * no scratch: a 1024-thread workgroup has 128 VGPRs per lane; a function call left in a kernel or
        addresses hoisted above an MLP chain push it into scratch spills;
      * every write-through output store written as inline asm is followed by its s_nop: a store wider
        than 64 bits still reads its data registers after issue, and the compiler cannot pad a hazard it
        does not see (without it results changed from run to run)."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = tmp_path / "kr.s"
    sys.path.insert(0, ROOT)
    from diffusionsat_amd.build import FLAGS      # the shipped library's flags
    subprocess.check_call(["hipcc"] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "diffusionsat_amd", "csrc", "kernels_round.hip")], stderr=subprocess.DEVNULL)
    text = out.read_text()
    import re
    kernels = re.findall(r"^(_ZN4dsat\S*round_p\d\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) >= 5                       # p1, p3, p2<1>, p2<2>, p2<4>
    for name, body in kernels:
        assert "scratch_" not in body, "%s uses scratch" % name
        lines = body.split("\n")
        stores = [i for i, l in enumerate(lines) if "global_store_dwordx4" in l and "sc0 sc1" in l]
        assert stores, name
        for i in stores:
            assert lines[i + 1].split(";")[0].strip().startswith("s_nop"), "%s: asm store without its s_nop" % name
    # and the resource records agree
    for m in re.finditer(r"\.name:\s+(\S*round_p\d\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text):
        assert int(m.group(2)) == 0, m.group(1)
