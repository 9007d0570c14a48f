"""Compact summary of a rocprofv3 --kernel-trace --stats run (kernel_stats.csv)."""
import csv, re, sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"# {path}\n# kernels: {len(rows)}  launches/step: {calls/steps:.0f}  GPU busy/step: {tot/steps/1e3:.0f} us  (steps={steps:.0f})")
print(f"{'kernel':60s} {'calls/step':>10s} {'avg_us':>9s} {'us/step':>9s} {'%':>6s}")
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n[:60]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    t = float(r["TotalDurationNs"])
    print(f"{short(r['Name']):60s} {int(r['Calls'])/steps:10.1f} {float(r['AverageNs'])/1e3:9.2f} {t/steps/1e3:9.1f} {100*t/tot:6.2f}")
