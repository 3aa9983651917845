"""YAML model-config loading + signature validation.
Mirror of training/caiman_asr_train/rnnt/config.py:31-75,136-142: the unchanged
training/configs/*.yaml of the reference load here, and unknown `rnnt:` keys are rejected
against the RNNT constructor signature."""
import inspect
from pathlib import Path
from typing import Dict

import yaml


def default_args(klass) -> Dict:
    sig = inspect.signature(klass.__init__)
    return {k: v.default for k, v in sig.parameters.items() if k != "self"}


def load(fpath) -> Dict:
    fpath = str(fpath)
    if fpath.endswith(".toml"):
        raise ValueError(".toml config format has been changed to .yaml")
    if Path(fpath).stat().st_size == 0:
        raise ValueError(f"Config file {fpath} is empty")
    cfg = yaml.safe_load(open(fpath, "r"))

    class _NoAlias(yaml.SafeDumper):
        def ignore_aliases(self, data):
            return True

    return yaml.safe_load(yaml.dump(cfg, Dumper=_NoAlias))  # deep-copies anchor-shared nodes


def validate_and_fill(klass, user_conf, ignore=(), optional=(), deprecated=()):
    conf = default_args(klass)
    to_ignore = set(ignore).union(deprecated)
    for k, v in user_conf.items():
        assert k in conf or k in to_ignore, f"Unknown parameter {k} for {klass}"
        if k in deprecated:
            continue
        conf[k] = v
    conf = {k: v for k, v in conf.items() if k not in optional or v is not inspect.Parameter.empty}
    for k, v in conf.items():
        assert v is not inspect.Parameter.empty, f"Value for {k} not specified for {klass}"
    return conf


def rnnt(conf) -> Dict:
    from caiman_asr_amd.rnnt.model import RNNT

    return validate_and_fill(RNNT, conf["rnnt"], optional=["n_classes"], deprecated=["hard_activation_functions"])
