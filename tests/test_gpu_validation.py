"""Input contracts the reference enforces by raising (nn.Embedding's IndexError, scatter's index checks, order-free
global_add_pool) and how the engine honours them: out-of-range indices are never dereferenced and surface as an
IndexError at the next check point; an unsorted ``batch`` vector gives the same read-out as a sorted one."""
import types

import pytest
import torch

from tests._util import TOL, assert_close, load, sub, tensors

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _clean():
    import phc_gnn_amd
    try:
        phc_gnn_amd.check_indices()
    except IndexError:
        pass


def test_out_of_range_edge_endpoint_is_reported_not_dereferenced():
    import phc_gnn_amd
    from phc_gnn_amd import ops
    from phc_gnn_amd.graph import EdgeStructure
    _clean()
    N, E, h = 50, 200, 32
    g = torch.Generator().manual_seed(1)
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, h, generator=g).to(DEV)
    for row, bad in ((0, N), (1, N + 7), (0, -1), (1, 1 << 40)):
        e = ei.clone()
        e[row, 17] = bad
        for pair in (True, False):                      # the 5-launch pair build and the general one
            st = EdgeStructure(e.to(DEV), N, N)
            if pair:
                st.prepare(for_backward=True)
            out = ops.aggregate_fwd_raw(x, None, st, "identity", True)
            _ = st.csc
            torch.cuda.synchronize()                    # no memory fault
            assert bool(torch.isfinite(out).all())
            with pytest.raises(IndexError, match="edge_index"):
                phc_gnn_amd.check_indices()
            phc_gnn_amd.check_indices()                 # sticky word was reset
    # the valid list raises nothing
    st = EdgeStructure(ei.to(DEV), N, N).prepare(for_backward=True)
    ops.aggregate_fwd_raw(x, None, st, "identity", True)
    phc_gnn_amd.check_indices()


def test_out_of_vocabulary_features_are_reported():
    import phc_gnn_amd
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    from phc_gnn_amd import harness
    _clean()
    model = harness.build_model("molhiv_b32", DEV, seed=0).train()
    gb = harness.make_batch("molhiv_b32", seed=1)
    d = harness.to_device(gb, DEV)
    model(d)
    phc_gnn_amd.check_indices()                          # clean batch: nothing
    d.x = d.x.clone()
    d.x[5, 1] = 4                                        # second atom feature has 4 values: 0..3
    out = model(d)
    assert bool(torch.isfinite(out).all())
    with pytest.raises(IndexError, match="node feature"):
        phc_gnn_amd.check_indices()
    d = harness.to_device(gb, DEV)
    d.edge_attr = d.edge_attr.clone()
    d.edge_attr[3, 2] = -1                               # third bond feature: 0..1
    out = model(d)
    out.sum().backward()
    assert bool(torch.isfinite(out).all())
    with pytest.raises(IndexError, match="edge feature"):
        phc_gnn_amd.check_indices()


def test_unsorted_batch_vector_gives_the_reference_read_out():
    """global_add_pool is a scatter in the reference: node order does not matter.  A data object that does not
    declare its ``batch`` vector sorted takes the grouping path; a permuted copy of the G8 batch must reproduce the
    golden logits."""
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    z = tensors(load("g8_model_cfg1"))
    model = PHMSkipConnectAdd(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                              downstream_layers=[128, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first",
                              msg_aggr="sum", norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", pooling="softattention")
    model.load_state_dict(sub(z, "sd::"), strict=True)
    model.to(DEV).train()
    N = z["x"].shape[0]
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(5))        # new position p holds old node perm[p]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(N)
    data = types.SimpleNamespace(x=z["x"][perm].to(DEV), edge_index=inv[z["edge_index"]].to(DEV),
                                 edge_attr=z["edge_attr"].to(DEV), batch=z["batch"][perm].to(DEV),
                                 num_graphs=int(z["y"].shape[0]))
    assert not bool((data.batch[1:] >= data.batch[:-1]).all())
    assert_close(model(data), z["logits_train"], 2e-5, "logits of the node-permuted batch")
