# Values of ogb 1.2.4; corroborated by the embedding-table shapes pickled in
# /root/reference/benchmarks/hiv/experiment1/run_1/model.pt.
def get_atom_feature_dims():
    return [119, 4, 12, 12, 10, 6, 6, 2, 2]


def get_bond_feature_dims():
    return [5, 6, 2]
