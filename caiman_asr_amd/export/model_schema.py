"""State-dict shape schemas of the supported model variants (training/caiman_asr_train/export/model_schema/
__init__.py:7-70).  The reference ships base.json / large.json; here the same tables are produced by instantiating
this package's RNNT with the base / large hyper-parameters on the meta device (no memory), so the schema cannot
drift from the model -- `tests/test_host_logic.py` pins both against the reference's JSON files."""
from enum import Enum
from typing import Dict, List

import torch

BASE = dict(in_feats=240, enc_n_hid=1024, enc_pre_rnn_layers=2, enc_post_rnn_layers=6, enc_stack_time_factor=2,
            pred_n_hid=512, pred_rnn_layers=2, joint_n_hid=768, n_classes=8704)
LARGE = dict(in_feats=240, enc_n_hid=1536, enc_pre_rnn_layers=2, enc_post_rnn_layers=6, enc_stack_time_factor=2,
             pred_n_hid=768, pred_rnn_layers=2, joint_n_hid=1024, n_classes=17408)


class CheckpointNotSupportedError(Exception):
    pass


class ModelVariant(Enum):
    BASE = "base"
    LARGE = "large"


def get_schema(state_dict: Dict) -> Dict[str, List[int]]:
    return {k: list(v.shape) for k, v in state_dict.items()}


def _variant_schema(hp: Dict) -> Dict[str, List[int]]:
    from caiman_asr_amd.rnnt.model import RNNT

    hp = dict(hp)
    n_classes = hp.pop("n_classes")
    with torch.device("meta"):
        m = RNNT(n_classes=n_classes, enc_batch_norm=False, pred_batch_norm=False, enc_dropout=0.0, pred_dropout=0.0,
                 joint_dropout=0.0, forget_gate_bias=1.0, weights_init_scale=1.0, hidden_hidden_bias_scale=0.0,
                 custom_lstm=False, **hp)
    return get_schema(m.state_dict())


_CACHE: Dict[str, Dict] = {}


def return_schemas() -> List[Dict[str, List[int]]]:
    for v, hp in ((ModelVariant.BASE, BASE), (ModelVariant.LARGE, LARGE)):
        if v.value not in _CACHE:
            _CACHE[v.value] = _variant_schema(hp)
    return [_CACHE[v.value] for v in ModelVariant]


def check_model_schema(model_sd: Dict, schemas=None) -> None:
    """Raise CheckpointNotSupportedError unless the state dict has exactly the names and shapes of one variant."""
    mine = get_schema(model_sd)
    if sum(1 for s in (schemas or return_schemas()) if s == mine) != 1:
        raise CheckpointNotSupportedError(
            "Model checkpoint's state dict sizes does not match any of the supported "
            f"ModelVariant options={[x.name for x in ModelVariant]}.")


def check_schema_training(model_sd: Dict, skip_state_dict_check: bool) -> None:
    try:
        check_model_schema(model_sd)
    except CheckpointNotSupportedError as e:
        if not skip_state_dict_check:
            raise CheckpointNotSupportedError(
                str(e) + "\nIf you would like to avoid this check, pass --skip_state_dict_check. NOTE that skipping "
                "this check will make your model incompatible with the Myrtle.ai inference server.")
