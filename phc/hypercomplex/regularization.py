"""Explicit L_p penalties added to the loss by the training scripts
(reference: phc/hypercomplex/regularization.py; called at
benchmarks/train_hiv.py:181-196).  Weights are located by ATTRIBUTE NAME
(``W`` / ``phm_rule``) on any sub-module, which is why those names are part of
the drop-in contract.  Plain PyTorch on whatever device the parameters live on."""


def multiplication_rule_regularization(model, p: int = 1, device=None):
    reg = 0.0
    for _, module in model.named_modules():
        rule = getattr(module, "phm_rule", None)
        if rule is not None:
            reg = reg + rule.norm(p=p).mean()
    return reg


def phm_weight_regularization(model, p: int = 2, device=None):
    weights = [module.W for _, module in model.named_modules() if getattr(module, "W", None) is not None]
    if weights and all(w.is_cuda for w in weights):
        # ROCm weights: the same value (and its gradient) from one engine launch instead of three small launches per layer each way
        from phc_gnn_amd import ops as _engine
        reg = _engine.weight_regularization(weights, p)
        if reg is not None:
            return reg
    reg = 0.0
    for weight in weights:
        reg = reg + weight.norm(p=p, dim=0).mean()
    return reg
