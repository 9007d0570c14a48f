"""The H-stationary GEMM's prologue loads its accumulator words and norm constants with inline-asm loads that hipcc's wait-count
insertion does not see, and waits for them with its own `s_waitcnt vmcnt(N)` (csrc/gemm_hs_device.h: hs_ld64 / hs_ld32 /
acc_arrived).  That is only correct while NO instruction reads or copies those registers between the loads and the counted wait.
This script compiles phm_linear.hip to gfx950 assembly and checks exactly that for every instantiated variant, plus the ring's
own counted waits and DMA instruction counts (audit_ring); phc_gnn_amd.build runs it whenever the file is recompiled and fails the
build on a violation, and tests/test_hs_asm_contract.py runs it too.   python scripts/audit_hs_asm_loads.py [existing.s]"""
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


def audit_ring(asm_text: str):
    """The LDS-DMA ring of every gemm_hs16_kernel instantiation: [(kernel, problems)].  What the generated code must still say
    for the hand-counted waits of csrc/gemm_hs_device.h to mean what they were written to mean:
      * the number of global_load_lds_dwordx4 instructions = (R - 1 + 2) * IPW * images — the ring's first fill plus ONE refill
        per step body (the peeled first step and the loop body): a compiler that duplicated or dropped a DMA would shift what
        `vmcnt(N)` counts;
      * the counted waits present are exactly the ones the source derives from (R, IPW, images, prologue loads);
      * plain / ST / BW variants: between a step's counted wait and its `s_barrier` no LDS read is issued (the wait covers only
        the wavefront's OWN pieces; the other wavefronts' pieces are valid behind the barrier only)."""
    out = []
    for m in re.finditer(r"^(_ZN3phc16gemm_hs16_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\w+):", asm_text, re.M):
        name = m.group(1)
        K, TPW, WAVES, RB, R, FLAGS = (int(m.group(i)) for i in range(2, 8))
        end = asm_text.index(".Lfunc_end", m.end())
        lines = [ln.strip() for ln in asm_text[m.end():end].split("\n")]
        XF, BW, BA = bool(FLAGS & 2), bool(FLAGS & 4), bool(FLAGS & 8)
        rpi = 256 // K if K in (128, 256) else 1
        ipw = (16 * RB // rpi) // WAVES
        imgs = 2 if BA else 1
        G = K // 16
        pro = TPW + TPW * G + (4 * TPW if BW else 0)
        vm_keep = (R - 2) * ipw * imgs
        vm_mid = (R - 2 - (1 if BW else 0)) * ipw * imgs
        acc_younger = (R - 1) * ipw * imgs
        problems = []
        n_dma = sum(1 for t in lines if t.startswith("global_load_lds_dwordx4"))
        if n_dma != (R - 1 + 2) * ipw * imgs:
            problems.append(f"{n_dma} LDS-DMA instructions, expected {(R - 1 + 2) * ipw * imgs}")
        in_asm, waits = False, []
        for n, t in enumerate(lines):
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm:
                w = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
                if w:
                    waits.append((n, int(w.group(1))))
        have = {v for _, v in waits}
        first = min(vm_keep + pro, 63)                    # (vmcnt is a 6-bit counter: the source saturates the first step's count)
        allowed = {0, first}
        need = {first}
        if XF or BA:
            allowed |= {vm_mid, acc_younger}
            need |= {vm_mid}
        else:
            allowed |= {vm_keep}
            need |= {vm_keep}
        if have - allowed:
            problems.append(f"unexpected counted waits {sorted(have - allowed)} (allowed {sorted(allowed)})")
        if need - have:
            problems.append(f"missing counted waits {sorted(need - have)}")
        if not (XF or BA):
            for n, v in waits:
                if v not in (vm_keep, first):
                    continue
                for t in lines[n + 1:]:
                    if t.startswith("s_barrier"):
                        break
                    if t.startswith("ds_read") or t.startswith("ds_load"):
                        problems.append(f"LDS read between the step's vmcnt({v}) wait and its barrier: {t}")
                        break
        out.append((name, problems))
    return out


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
    ring = audit_ring(text)
    ring_bad = 0
    for name, problems in ring:
        for pr in problems:
            print(f"{name[:70]:70s} RING: {pr}")
            ring_bad += 1
    print(f"{len(ring)} instantiations, {ring_bad} ring problems")
    return 1 if worst or not res or ring_bad or not ring else 0


if __name__ == "__main__":
    sys.exit(main())
