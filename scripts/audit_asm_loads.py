"""Audit hipcc output for registers of inline-asm loads being touched by compiler code before
the inline-asm wait that retires them (cdna guide §5.7 item 1).  Linear scan per kernel."""
import re, sys

path, pat = sys.argv[1], sys.argv[2]
text = open(path).read()
kernels = re.split(r"\n(?=_ZN[A-Za-z0-9_]*:[^\n]*\n)", text)
bad_total = 0
for kern in kernels:
    name = kern.split(":", 1)[0].strip()
    if pat not in name:
        continue
    lines = kern.split("\n")
    pending, in_asm, bad = set(), False, 0
    scratch = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", kern)
    for ln, line in enumerate(lines):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm:
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", t)
            if m:
                pending |= set(range(int(m.group(1)), int(m.group(2)) + 1))
            if t.startswith("s_waitcnt vmcnt"):
                pending.clear()
            continue
        if not pending or not t or t.startswith(";") or t.startswith("."):
            continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
            regs |= set(range(int(a), int(b) + 1))
        for a in re.findall(r"\bv(\d+)\b", t):
            regs.add(int(a))
        hit = regs & pending
        if hit:
            bad += 1
            if bad <= 5:
                print(f"  !! {name[:60]} line {ln}: {t}   (pending {sorted(hit)[:4]}...)")
    print(f"{name[:90]}: scratch={scratch.group(1) if scratch else '?'} violations={bad}")
    bad_total += bad
sys.exit(1 if bad_total else 0)
