"""The H-stationary GEMM's prologue waits for its inline-asm accumulator loads with a counted `s_waitcnt` of its own
(csrc/gemm_hs_device.h: hs_ld64 / hs_ld32 / acc_arrived).  hipcc does not know those registers are pending, so the contract "no
instruction reads or copies them before the counted wait" can only be checked in the generated code: compile phm_linear.hip to
gfx950 assembly (no GPU needed, about a minute) and audit every instantiated variant (scripts/audit_hs_asm_loads.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_asm_loaded_registers_are_untouched_until_their_counted_wait(tmp_path):
    import audit_hs_asm_loads as audit
    path = str(tmp_path / "phm_linear.s")
    audit.compile_asm(path)
    results = audit.audit(open(path).read())
    # XF (2, 3), BA (8, 12) x the four (K, wavefronts) configurations
    assert len(results) >= 16, [r[0] for r in results]
    for name, n_loads, wait, bad in results:
        assert n_loads in (18, 20), (name, n_loads)                 # 16 accumulator words + 2 (XF: weight, bias) or 4 (BA: save_stat)
        assert wait.startswith("s_waitcnt vmcnt("), (name, wait)
        assert not bad, (name, bad[:4])
