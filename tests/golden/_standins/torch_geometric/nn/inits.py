def reset(module):
    if hasattr(module, "reset_parameters") and not hasattr(module, "children"):
        module.reset_parameters()
        return
    kids = list(module.children()) if hasattr(module, "children") else []
    for m in kids:
        if hasattr(m, "reset_parameters"):
            m.reset_parameters()
        else:
            reset(m)
