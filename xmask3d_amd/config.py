"""yaml -> flat attribute dict, plus ``key value`` command-line overrides.

Same contract as the reference's /root/reference/util/config.py:58-146: every top-level yaml section is
flattened into ONE namespace (later sections win), nested dicts stay attribute dicts, CLI pairs are
``literal_eval``-ed and must match the type of the value they replace (list<->tuple coerced, a ``None``
original accepts anything, dotted keys address by their last component).  Pinned by
tests/golden/config_b15n4.json (dump of the reference loader on its own B15N4 file + overrides).
"""
from __future__ import annotations

import ast
import copy
import os

import yaml


class CfgNode(dict):
    """dict with attribute access; nested dicts are converted recursively."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __str__(self):
        lines = []
        for k in sorted(self):
            v = self[k]
            if isinstance(v, CfgNode):
                lines.append(f"{k}:")
                lines += ["  " + l for l in str(v).split("\n")]
            else:
                lines.append(f"{k}: {v}")
        return "\n".join(lines)


def load_cfg_from_cfg_file(path):
    assert os.path.isfile(path) and path.endswith(".yaml"), "{} is not a yaml file".format(path)
    with open(path, "r") as f:
        sections = yaml.safe_load(f)
    flat = {}
    for section in sections.values():
        flat.update(section)
    return CfgNode(flat)


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    if old is None or type(new) is type(old):
        return new
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
        type(old), type(new), old, new, key))


def merge_cfg_from_list(cfg, pairs):
    assert len(pairs) % 2 == 0
    out = copy.deepcopy(cfg)
    for full_key, raw in zip(pairs[0::2], pairs[1::2]):
        key = full_key.split(".")[-1]
        assert key in cfg, "Non-existent key: {}".format(full_key)
        out[key] = _coerce(_decode(raw), cfg[key], full_key)
    return out
