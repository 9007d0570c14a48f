"""Optimizer parameter-group helper used by every training script
(reference: phc/quaternion/regularization.py:4-24; call sites
benchmarks/train_hiv.py:266-275)."""


def get_model_blocks(model, attr: str, **kwargs) -> list:
    """[{'params': [...], **kwargs}] for sub-module ``attr`` of ``model`` (or of
    ``model.module`` when wrapped), [] when absent or parameter-free."""
    owner = getattr(model, "module", model)
    if not hasattr(owner, attr):
        owner = model
    block = getattr(owner, attr, None)
    if block is None:
        return []
    params = list(block.parameters())
    return [dict(params=params, **kwargs)] if params else []


def quaternion_weight_regularization(model, device, p: int = 1):
    raise NotImplementedError("the quaternion (QTensor) stack is outside this engine's scope; "
                              "PHM with c_init='standard', phm_dim=4 is the quaternion layer")
