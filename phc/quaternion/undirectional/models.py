"""Import-compatibility stubs: the reference's training scripts import these
names unconditionally (benchmarks/train_hiv.py:27-28).  The fixed-algebra
quaternion stack is out of scope; PHM with
c_init='standard', phm_dim=4 computes the same layer."""


class _OutOfScope:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}: the QTensor quaternion stack is not part of phc_gnn_amd; "
                                  "use --type undirectional-phm-sc-add with phm_dim=4, c_init=standard")


class QuaternionSkipConnectAdd(_OutOfScope):
    pass


class QuaternionSkipConnectConcat(_OutOfScope):
    pass
