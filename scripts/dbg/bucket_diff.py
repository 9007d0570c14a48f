"""Which parameter gradients move when the same batch is padded to a larger bucket?  (bit-level diff per tensor)"""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_bucket_independence as T
from phc_gnn_amd import graphed, harness
for workload, dn, de in [("molhiv_b512", 256, 0), ("molhiv_b512", 0, 256), ("molhiv_b512", 256, 128), ("zinc_b1024", 512, 0), ("zinc_b1024", 0, 512),
                         ("molhiv_b512_softmax", 256, 0), ("molhiv_b512_softmax", 0, 256), ("molpcba_b1024", 512, 0)]:
    gb = harness.make_batch(workload, seed=21)
    n_max, e_max = graphed.bucket_shape([gb])
    la, ga = T._grads(workload, gb, n_max, e_max)
    lb, gb_ = T._grads(workload, gb, n_max + dn, e_max + de)
    bad = [(k, float((ga[k] - gb_[k]).abs().max() / max(float(ga[k].abs().max()), 1e-30))) for k in ga if not torch.equal(ga[k], gb_[k])]
    print(workload, (n_max, e_max), "+", (dn, de), "loss equal:", la == lb, "differing:", len(bad), "of", len(ga))
    for k, e in bad:
        print("    ", k, "%.2e" % e)
