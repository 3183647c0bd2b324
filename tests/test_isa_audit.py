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


@pytest.mark.parametrize("src", ["kernels_round.hip", "kernels_chain.hip"])
def test_no_instruction_touches_a_register_with_a_load_in_flight(src):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_ring.py"), src],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert "asm loads" in res.stdout and "hazards 0" in res.stdout
