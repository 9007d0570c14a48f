"""Ordered kernel list of the LAST hipGraph replay in a rocprofv3 --kernel-trace csv:
start offset, duration and the idle gap before each launch."""
import csv, re, sys

path = sys.argv[1]
window_us = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= end - window_us * 1e3]
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n) if not n.startswith("(") else n
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n[:70]
t0 = int(sel[0]["Start_Timestamp"]); prev = t0; busy = 0
print(f"# {len(sel)} launches in the last {window_us:.0f} us")
for i, r in enumerate(sel):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:4d} t={(s-t0)/1e3:8.1f} dur={(e-s)/1e3:6.2f} gap={(s-prev)/1e3:6.2f}  {short(r['Kernel_Name'])}  grid={r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')} wg={r.get('Workgroup_Size_X','')} lds={r.get('LDS_Block_Size','')} vgpr={r.get('VGPR_Count','')}")
    prev = e; busy += e - s
print(f"# busy {busy/1e3:.1f} us, span {(prev-t0)/1e3:.1f} us")
