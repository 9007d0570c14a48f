"""The H-stationary GEMM's prologue loads its accumulator words and norm constants with inline-asm loads that hipcc's wait-count
insertion does not see, and waits for them with its own `s_waitcnt vmcnt(N)` (csrc/gemm_hs_device.h: hs_ld64 / hs_ld32 /
acc_arrived).  That is only correct while NO instruction reads or copies those registers between the loads and the counted wait.
This script compiles phm_linear.hip to gfx950 assembly and checks exactly that for every instantiated variant; it is also what
tests/test_hs_asm_contract.py runs.   python scripts/audit_hs_asm_loads.py [existing.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(out_path: str) -> None:
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "phc_gnn_amd", "csrc", "phm_linear.hip")
    cmd = [hipcc, "-S", "--cuda-device-only", "-x", "hip", src, "-o", out_path, "-O3", "-std=c++17", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "phc_gnn_amd", "csrc"), "-w"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def audit(asm_text: str):
    """[(kernel, n_asm_loads, wait_instruction, [offending lines])] for every gemm_hs16_kernel instantiation with asm loads."""
    results = []
    for m in re.finditer(r"^(_ZN3phc16gemm_hs16_kernelI\w+):", asm_text, re.M):
        name = m.group(1)
        end = asm_text.index(".Lfunc_end", m.end())
        lines = asm_text[m.end():end].split("\n")
        in_asm, loads, wait_at = False, [], None
        for n, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and re.match(r"global_load_dword(x2)?\s", t) and "lds" not in t:
                loads.append((n, _regs(t.split(",")[0])))
            elif in_asm and t.startswith("s_waitcnt vmcnt") and loads and wait_at is None:
                wait_at = n
        if not loads:
            continue
        if wait_at is None:
            results.append((name, len(loads), "NO COUNTED WAIT", ["no counted wait behind the asm loads"]))
            continue
        bad = []
        for n in range(loads[0][0] + 1, wait_at):
            t = lines[n].strip().split(";")[0]
            if not t or t.startswith(".") or re.match(r"global_load_dword(x2)?\s", t) and any(n == k for k, _ in loads):
                continue
            loaded = set().union(*[rg for k, rg in loads if k < n])
            if _regs(t) & loaded:
                bad.append(t)
        results.append((name, len(loads), lines[wait_at].strip(), bad))
    return results


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "phm_linear.s")
            compile_asm(path)
            text = open(path).read()
    res = audit(text)
    worst = 0
    for name, n, wait, bad in res:
        print(f"{name[:70]:70s} asm loads {n:2d}  {wait:22s} touched before the wait: {len(bad)}")
        for b in bad[:4]:
            print("      ", b)
        worst += len(bad)
    print(f"{len(res)} variants with asm loads, {worst} offending instructions")
    return 1 if worst or not res else 0


if __name__ == "__main__":
    sys.exit(main())
