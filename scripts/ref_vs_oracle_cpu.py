#!/usr/bin/env python3
"""Container-only check of SURVEY §8(d)(i): the CPU oracle (oracle/phc_oracle.py, what bench.py's `cpu_baseline` times on
the GPU box) must cost what the REFERENCE costs on the same host, threads and batch — otherwise the GPU/CPU ratio in the
bench line would be quoted against a straw man.  Imports the real reference from /root/reference (with the stand-ins of
tests/golden/_standins for torch-scatter / PyG / ogb), builds BASELINE configs[1] (molhiv PHM-4, hidden 128, 4 layers,
512 graphs), and times the reference loop body (train_hiv.py:173-201: zero_grad, forward, masked BCE, backward,
clip_grad_norm_(2.0), Adam) next to the same step through the oracle.

    python scripts/ref_vs_oracle_cpu.py [--threads 6] [--steps 20]        # prints one JSON line; see BASELINE.md
"""
import argparse
import importlib.util
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)          # the reference's torch.set_num_threads(6), train_hiv.py:632
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("needs the reference checkout at /root/reference (build container only)")
    import numpy as np
    import torch
    torch.set_num_threads(args.threads)

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    synth = load(os.path.join(REPO, "phc_gnn_amd", "synth.py"), "phc_synth")
    oracle = load(os.path.join(REPO, "oracle", "phc_oracle.py"), "phc_oracle_standalone")
    sys.path.insert(0, os.path.join(REPO, "tests", "golden", "_standins"))
    sys.path.insert(0, REF)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != REPO]
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    from torch_geometric.data import Batch

    gb = synth.molecule_batch(512, seed=0)
    kw = dict(phm_dim=4, learn_phm=True, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
              msg_aggr="sum", norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", pooling="softattention", activation="relu",
              real_trafo="linear", w_init="phm", c_init="standard", downstream_layers=[128, 32], dropout_dn=[0.0, 0.0],
              target_dim=1, sc_type="first")
    torch.manual_seed(0)
    np.random.seed(0)
    model = PHMSkipConnectAdd(**kw).train()
    with torch.no_grad():
        for m in model.modules():                                  # the reference leaves b[out/n] uninitialised (layers.py:276-278)
            if hasattr(m, "b") and isinstance(getattr(m, "b"), torch.nn.Parameter):
                m.b.nan_to_num_(0.0).clamp_(-1, 1)
    data = Batch(x=torch.from_numpy(gb.x), edge_index=torch.from_numpy(gb.edge_index), edge_attr=torch.from_numpy(gb.edge_attr),
                 batch=torch.from_numpy(gb.batch), y=torch.from_numpy(gb.y), num_graphs=gb.num_graphs)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def timed(step):
        for _ in range(3):
            step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        return (time.perf_counter() - t0) / args.steps

    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def ref_step():
        opt.zero_grad()
        logits = model(data)
        y = data.y.to(torch.float)
        mask = ~torch.isnan(y)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits[mask], y[mask])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=2.0, norm_type=2)
        opt.step()
    t_ref = timed(ref_step)

    names = [k for k, _ in model.named_parameters()]
    p = {k: v.clone() for k, v in sd.items()}
    for k in names:
        p[k].requires_grad_()
    params = [p[k] for k in names]
    opt2 = torch.optim.Adam(params, lr=1e-3)
    x, ei, ea, batch, y = data.x, data.edge_index, data.edge_attr, data.batch, data.y

    def oracle_step():
        opt2.zero_grad(set_to_none=True)
        logits = oracle.skip_connect_add_forward(p, x, ei, ea, batch, phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3,
                                                 num_dn_hidden=2, mlp=True, sc_type="first", training=True, num_graphs=gb.num_graphs)
        loss = oracle.bce_loss_masked(logits, y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm=2.0)
        opt2.step()
    t_or = timed(oracle_step)
    print(json.dumps({"threads": args.threads, "steps": args.steps, "graphs_per_batch": gb.num_graphs, "nodes": gb.num_nodes,
                      "edges": gb.num_edges, "reference_ms_per_step": round(1e3 * t_ref, 2), "oracle_ms_per_step": round(1e3 * t_or, 2),
                      "reference_graphs_per_s": round(gb.num_graphs / t_ref, 1), "oracle_graphs_per_s": round(gb.num_graphs / t_or, 1),
                      "oracle_over_reference_time": round(t_or / t_ref, 3), "host_cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()
