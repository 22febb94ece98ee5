"""Config helpers with the semantics of the reference's vqa/lib/utils.py:6-12 (CLI-over-YAML merge)."""


def update_values(dict_from, dict_to):
    """Recursively copy values of `dict_from` into `dict_to`; a value of None does NOT override
    (that is how unset argparse flags leave the YAML value alone: counterexamples.py:100-115)."""
    for key, value in dict_from.items():
        if isinstance(value, dict):
            update_values(value, dict_to.setdefault(key, {}))
        elif value is not None:
            dict_to[key] = value
    return dict_to


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v is None:
        return None
    s = str(v).lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise ValueError("boolean value expected, got %r" % (v,))


def params_count(model):
    return sum(p.numel() for p in model.parameters())
