"""CPU: the static scan of the library's gfx950 assembly for the hazards hipcc does not know about around hand-written instructions
(tools/audit_asm_waits.py): an asm LDS read's registers used before their wait, VALU writes around a wide asm LDS store's data
registers, asm VALU readers of an MFMA result inside its wait states.  Each of the three was once a wrong-result bug found on the
GPU; the scan needs no GPU (hipcc cross-compiles to assembly: about a minute)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_no_unprotected_asm_hazard_in_the_library():
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "audit_asm_waits.py")], capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (out.stdout + out.stderr)[-3000:]
    if out.returncode == 77:
        pytest.skip("no hipcc with gfx950 support on this box: " + tail.strip()[-200:])
    assert out.returncode == 0, tail
    # the scan must actually have seen the kernels (a label pattern that matches nothing reports zero findings too)
    assert "kernels scanned:" in out.stdout and int(out.stdout.split("kernels scanned:")[1].split()[0]) >= 100, tail
