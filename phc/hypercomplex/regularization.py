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
    reg = 0.0
    for _, module in model.named_modules():
        weight = getattr(module, "W", None)
        if weight is not None:
            reg = reg + weight.norm(p=p, dim=0).mean()
    return reg
