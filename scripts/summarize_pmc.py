"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units), merged into a JSON file that
is keyed by WORKLOAD (bench.py's pmc_traffic reads it):  summarize_pmc.py <fetch csv> <write csv> <json> <workload>
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of a wide coalesced read stream."""
import csv, re, sys, json, collections, os

def load(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc

f, w = load(sys.argv[1]), load(sys.argv[2])
workload = sys.argv[4] if len(sys.argv) > 4 else "molhiv_b512"
out = {}
print(f"# workload {workload}")
print(f"{'kernel':58s} {'launches':>8s} {'fetch KiB/launch (raw)':>24s} {'write KiB/launch':>18s} {'HBM bytes/launch (fetch x2 + write)':>36s}")
for k in sorted(f, key=lambda k: -f[k][0]):
    if "phc::" not in k:
        continue
    fl = f[k][0] / max(f[k][1], 1)
    wl = w[k][0] / max(w[k][1], 1) if k in w else 0.0
    hbm = (2 * fl + wl) * 1024
    out[k] = {"launches": f[k][1], "fetch_kib_raw": round(fl, 1), "write_kib": round(wl, 1), "hbm_bytes_per_launch": round(hbm)}
    print(f"{k[:58]:58s} {f[k][1]:8d} {fl:24.1f} {wl:18.1f} {hbm:36.0f}")
table = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
table[workload] = out
json.dump(table, open(sys.argv[3], "w"), indent=1)
