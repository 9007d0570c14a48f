import sys, torch
sys.path.insert(0, "tests")
from phc_gnn_amd import graphed, harness
import test_gpu_bucket_independence as T
wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b512"
gb = harness.make_batch(wl, seed=21)
n_max, e_max = graphed.bucket_shape([gb])
print("N", gb.num_nodes, "E", gb.num_edges, "bucket", n_max, e_max)
la, ga = T._grads(wl, gb, n_max, e_max)
for en, ee in ((0, 0), (128, 0), (0, 128), (512, 256), (1024, 512), (2048, 1024), (4096, 2048), (4224, 2048)):
    lb, gb_ = T._grads(wl, gb, n_max + en, e_max + ee)
    rows = []
    for k in ga:
        if not torch.equal(ga[k], gb_[k]):
            rows.append((float((ga[k] - gb_[k]).abs().max()) / max(float(ga[k].abs().max()), 1e-30), k, float(ga[k].abs().max())))
    rows.sort(reverse=True)
    print(f"+{en} nodes +{ee} edges: loss equal {la == lb}; {len(rows)} of {len(ga)} differ; worst:", [(f"{r[0]:.2e}", r[1], f"{r[2]:.2e}") for r in rows[:4]])
