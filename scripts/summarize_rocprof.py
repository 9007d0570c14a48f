"""Compact per-step summary of a rocprofv3 --kernel-trace --stats run.

    summarize_rocprof.py <kernel_stats.csv | kernel_trace.csv> [steps] [top] [marker substring]

Given the kernel TRACE, the table is restricted to the REPLAY WINDOW: the last `steps` whole training steps of the run, a step
being the span from one launch of the step's last kernel (the marker, `opt_adam_kernel` when present, otherwise the most frequent
periodic kernel) to the next.  Set-up launches (model construction, warm-up passes, the eager profile pass, capture) are outside
the window, so `calls/step`, `us/step` and `%` are per replayed step.  Given only the kernel STATS file (no timestamps) the table
is what it was before round 5: totals over the whole process divided by `steps` — diluted by set-up launches, and says so."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n) if not n.startswith("(") else n
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n[:60]


def table(rows_by_name, steps, top, header):
    tot = sum(t for _, t in rows_by_name.values())
    calls = sum(c for c, _ in rows_by_name.values())
    print(header)
    print(f"# kernels: {len(rows_by_name)}  launches/step: {calls / steps:.1f}  GPU busy/step: {tot / steps / 1e3:.1f} us  (steps={steps:.0f})")
    print(f"{'kernel':60s} {'calls/step':>10s} {'avg_us':>9s} {'us/step':>9s} {'%':>6s}")
    for name, (c, t) in sorted(rows_by_name.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{short(name):60s} {c / steps:10.1f} {t / c / 1e3:9.2f} {t / steps / 1e3:9.1f} {100 * t / tot:6.2f}")


def from_trace(path, steps, top, marker_sub=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    if marker_sub:
        marker = next((n for n in set(names) if marker_sub in n), None)
    else:
        marker = next((n for n in set(names) if ("opt_adam_kernel" in n or "opt_clip_adam_fused_kernel" in n)), None)
    if marker is None:
        cnt = defaultdict(int)
        for n in names:
            cnt[n] += 1
        marker = max(cnt, key=cnt.get)
    ends = [i for i, n in enumerate(names) if n == marker]
    if len(ends) < 2:
        sys.exit(f"{path}: fewer than two launches of the step marker {short(marker)}")
    steps = int(min(steps, len(ends) - 1))
    lo, hi = ends[-1 - steps] + 1, ends[-1] + 1                  # launches (lo .. hi) = the last `steps` whole steps
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])
    table({k: tuple(v) for k, v in agg.items()}, float(steps), top,
          f"# {path}\n# replay window: the last {steps} steps (marker {short(marker)}); period {span / steps / 1e3:.1f} us/step; "
          f"{lo} earlier launches (set-up, warm-up) excluded")


def from_stats(path, steps, top):
    rows = list(csv.DictReader(open(path)))
    table({r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows}, steps, top,
          f"# {path}\n# WHOLE PROCESS totals / steps (no timestamps in a stats file): set-up launches dilute calls/step and %")


if __name__ == "__main__":
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    head = open(path).readline()
    if "Start_Timestamp" in head:
        from_trace(path, steps, top, sys.argv[4] if len(sys.argv) > 4 else None)     # [4]: substring of the step's marker kernel
    else:
        from_stats(path, steps, top)
