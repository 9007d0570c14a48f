"""Per-kernel MFMA utilisation from rocprofv3 --pmc passes (scripts/rocprof_pmc_mfma.sh), merged into a JSON file keyed by WORKLOAD
(bench.py's pmc_mfma reads it):   summarize_pmc_mfma.py <json> <workload> <counter csv of pass A> [<counter csv of pass B>]

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)
      SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over the chip (MI355X_MICROARCH.md: = 32 x the number of
      v_mfma_f32_16x16x4_f32 issued, 64 x for 32x32x2);  SQ_BUSY_CU_CYCLES: cycles a CU has any wavefront, summed over the CUs.
  mfma_busy_chip = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
      the same against the whole chip for the whole dispatch (GRBM_GUI_ACTIVE is summed over the 8 XCDs): a launch that leaves CUs idle
      or spends its time in prologue / epilogue shows it here, not in the first figure.
Dispatches are matched per kernel NAME (every instantiation its own row); values are means over the dispatches of the pass."""
import collections
import csv
import json
import os
import re
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        c = acc[name][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
    return acc


out_json, workload, pa = sys.argv[1], sys.argv[2], sys.argv[3]
pb = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else None
A = load(pa)
B = load(pb) if pb and os.path.exists(pb) else {}
rows = {}
print(f"# workload {workload}")
print(f"{'kernel':62s} {'launches':>8s} {'MFMA busy cyc':>14s} {'CU busy cyc':>13s} {'GUI_ACTIVE/8':>13s} {'mfma_busy':>10s} {'mfma_busy_chip':>15s} {'MFMA MOPS f32':>14s}")


def mean(d, k):
    return d[k][0] / d[k][1] if k in d and d[k][1] else None


for name in sorted(A, key=lambda k: -(mean(A[k], "SQ_VALU_MFMA_BUSY_CYCLES") or 0) * (A[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1] or 1)):
    if "phc::" not in name:
        continue
    mf, cu, gui = mean(A[name], "SQ_VALU_MFMA_BUSY_CYCLES"), mean(A[name], "SQ_BUSY_CU_CYCLES"), mean(A[name], "GRBM_GUI_ACTIVE")
    n = A[name]["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    if not mf:
        continue
    busy = mf / (4.0 * cu) if cu else None
    chip = mf / (1024.0 * gui / 8.0) if gui else None
    mops = mean(B.get(name, {}), "SQ_INSTS_VALU_MFMA_MOPS_F32") if B else None
    rows[name] = {"launches": n, "mfma_busy_cycles": round(mf), "cu_busy_cycles": round(cu) if cu else None,
                  "gui_active_per_xcd": round(gui / 8.0) if gui else None, "mfma_busy": round(busy, 4) if busy is not None else None,
                  "mfma_busy_chip": round(chip, 4) if chip is not None else None, "mfma_mops_f32": round(mops) if mops else None}
    print(f"{name[:62]:62s} {n:8d} {mf:14.0f} {cu or 0:13.0f} {(gui or 0) / 8:13.0f} {busy if busy is not None else float('nan'):10.4f} "
          f"{chip if chip is not None else float('nan'):15.4f} {mops or 0:14.0f}")
table = json.load(open(out_json)) if os.path.exists(out_json) else {}
table[workload] = rows
json.dump(table, open(out_json, "w"), indent=1)
