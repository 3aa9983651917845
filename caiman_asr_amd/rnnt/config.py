"""YAML model-config loading + validation against a constructor signature.

Surface of training/caiman_asr_train/rnnt/config.py:31-75,136-142 (`default_args`, `load`, `validate_and_fill`,
`rnnt`): the unchanged training/configs/*.yaml of the reference load here, anchors are expanded into independent
copies, and a key of the `rnnt:` section that the RNNT constructor does not take is rejected with the reference's
wording ("Unknown parameter ...", "Value for ... not specified ...").  The bodies are this repo's own: the signature
is bound once into a table of (name, default) pairs, the user keys are classified against it in one pass.
"""
import copy
import inspect
import os
from typing import Dict, Iterable

import yaml

_REQUIRED = inspect.Parameter.empty


def _signature_table(klass):
    """[(keyword, default | _REQUIRED)] of klass.__init__, `self` and *args / **kwargs catch-alls left out."""
    table = []
    for name, par in list(inspect.signature(klass.__init__).parameters.items())[1:]:
        if par.kind in (inspect.Parameter.VAR_POSITIONAL, inspect.Parameter.VAR_KEYWORD):
            continue
        table.append((name, par.default))
    return table


def default_args(klass) -> Dict:
    return dict(_signature_table(klass))


def load(fpath) -> Dict:
    path = os.fspath(fpath)
    if os.path.splitext(path)[1] == ".toml":
        raise ValueError(f"{path}: the .toml config format was replaced by .yaml")
    if os.path.getsize(path) == 0:
        raise ValueError(f"{path}: config file is empty")
    with open(path, "r") as fh:
        tree = yaml.safe_load(fh)
    # YAML anchors (`&x` / `*x`, `<<: *x`) load as ONE shared object per anchor: give every user its own copy, so that
    # editing cfg["a"] after loading can never leak into cfg["b"]
    return copy.deepcopy(tree, memo=_NoSharing())


class _NoSharing(dict):
    """deepcopy memo that forgets: a node reached twice is copied twice."""

    def __setitem__(self, key, value):
        pass

    def get(self, key, default=None):
        return default


def validate_and_fill(klass, user_conf, ignore: Iterable[str] = (), optional: Iterable[str] = (),
                      deprecated: Iterable[str] = ()) -> Dict:
    """keyword dict for klass(**...): constructor defaults overlaid with `user_conf`.  Keys outside the signature must
    be listed in `ignore` / `deprecated` (deprecated keys are dropped); a parameter without default that the user did
    not give is an error unless it is `optional` (then it is left for the caller to supply)."""
    table = _signature_table(klass)
    known = {name for name, _ in table}
    passthrough, dropped = set(ignore), set(deprecated)
    strays = [k for k in user_conf if k not in known and k not in passthrough and k not in dropped]
    assert not strays, f"Unknown parameter {strays[0]} for {klass}"
    out = {}
    for name, default in table:
        value = user_conf[name] if (name in user_conf and name not in dropped) else default
        if value is _REQUIRED:
            assert name in optional, f"Value for {name} not specified for {klass}"
            continue
        out[name] = value
    for k in user_conf:           # keys the caller asked to carry along although the constructor does not take them
        if k in passthrough and k not in dropped and k not in known:
            out[k] = user_conf[k]
    return out


def rnnt(conf) -> Dict:
    from caiman_asr_amd.rnnt.model import RNNT

    return validate_and_fill(RNNT, conf["rnnt"], optional=["n_classes"], deprecated=["hard_activation_functions"])
